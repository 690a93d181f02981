/* main_tvl1.c — `tvl1flow`, drop-in for the reference's optical-flow tool
 * (reference: lib/tvl1flow/main.c): same positional arguments, defaults, parameter
 * checks and messages; the flow is written by extension (.flo / .tif / .pfm) with two
 * interleaved channels like iio_write_image_float_split does (main.c:177).
 *
 *   tvl1flow I0 I1 [out nproc tau lambda theta nscales fscale zfactor nwarps epsilon verbose]
 *
 * nproc is accepted and ignored (there are no host threads to configure). Colour inputs are
 * reduced to luminance the way the reference's reader does for float images
 * (lib/iio/iio.c:1048-1056); 8/16-bit colour files, which that reader truncates to integers,
 * are not what the pipelines feed it (float TIFF frames) and are converted without truncation. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cli_server.h"
#include "imgio.h"
#include "nlk_hip.h"

nlk_ctx *nlkalman_hip_context(void); /* libnlkalman.so: the process-wide device context */

static int fail_hip(const char *what, nlk_ctx *c) {
  fprintf(stderr, "tvl1flow (hip): %s: %s\n", what, nlk_last_error(c));
  return EXIT_FAILURE;
}

/* the tool as a function: main() below, or the resident server (main_server.c) */
static int tool_body(int argc, const char **argv);

/* (every way out of the body releases the host images: a resident process must not grow on failed requests) */
int nlk_tool_tvl1(int argc, const char **argv) {
  const int rc = tool_body(argc, argv);
  cli_host_release();
  return rc;
}

static int tool_body(int argc, const char **argv) {
  if (argc < 3) {
    fprintf(stderr, "Usage: %s I0 I1 [out "
                    "nproc tau lambda theta nscales fscale zfactor nwarps epsilon "
                    "verbose]\n", *argv);
    return EXIT_FAILURE;
  }
  struct nlk_tvl1_params D;
  nlk_tvl1_default_params(&D);
  int i = 1;
  const char *image1_name = argv[i++];
  const char *image2_name = argv[i++];
  const char *outfile = (argc > i) ? argv[i] : "flow.flo"; i++;
  int nproc = (argc > i) ? atoi(argv[i]) : 0; i++;
  float tau = (argc > i) ? atof(argv[i]) : D.tau; i++;
  float lambda = (argc > i) ? atof(argv[i]) : D.lambda; i++;
  float theta = (argc > i) ? atof(argv[i]) : D.theta; i++;
  int nscales = (argc > i) ? atoi(argv[i]) : D.nscales; i++;
  int fscale = (argc > i) ? atoi(argv[i]) : D.fscale; i++;
  float zfactor = (argc > i) ? atof(argv[i]) : D.zfactor; i++;
  int nwarps = (argc > i) ? atoi(argv[i]) : D.nwarps; i++;
  float epsilon = (argc > i) ? atof(argv[i]) : D.epsilon; i++;
  int verbose = (argc > i) ? atoi(argv[i]) : 0; i++;

  /* out-of-range values fall back to the defaults (reference: main.c:98-137) */
  if (nproc < 0) { nproc = 0; if (verbose) fprintf(stderr, "warning: nproc changed to %d\n", nproc); }
  if (tau <= 0 || tau > 0.25) { tau = D.tau; if (verbose) fprintf(stderr, "warning: tau changed to %g\n", tau); }
  if (lambda <= 0) { lambda = D.lambda; if (verbose) fprintf(stderr, "warning: lambda changed to %g\n", lambda); }
  if (theta <= 0) { theta = D.theta; if (verbose) fprintf(stderr, "warning: theta changed to %g\n", theta); }
  if (nscales <= 0) { nscales = D.nscales; if (verbose) fprintf(stderr, "warning: nscales changed to %d\n", nscales); }
  if (zfactor <= 0 || zfactor >= 1) { zfactor = D.zfactor; if (verbose) fprintf(stderr, "warning: zfactor changed to %g\n", zfactor); }
  if (nwarps <= 0) { nwarps = D.nwarps; if (verbose) fprintf(stderr, "warning: nwarps changed to %d\n", nwarps); }
  if (epsilon <= 0) { epsilon = D.epsilon; if (verbose) fprintf(stderr, "warning: epsilon changed to %f\n", epsilon); }

  int nx, ny, c0, nx2, ny2, c1;
  float *I0 = (float *)cli_host_keep(img_read(image1_name, &nx, &ny, &c0));
  if (!I0) { fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", image1_name); return EXIT_FAILURE; }
  float *I1 = (float *)cli_host_keep(img_read(image2_name, &nx2, &ny2, &c1));
  if (!I1) { fprintf(stderr, "ERROR: could not read image from file \"%s\"\n", image2_name); return EXIT_FAILURE; }
  if (nx != nx2 || ny != ny2) {
    fprintf(stderr, "ERROR: input images size mismatch %dx%d != %dx%d\n", nx, ny, nx2, ny2);
    return EXIT_FAILURE;
  }
  if (c0 == 2 || c0 > 4 || c1 == 2 || c1 > 4) {
    fprintf(stderr, "ERROR: non-scalarizable image\n"); /* reference: lib/iio/iio.c:3997-3998 */
    return EXIT_FAILURE;
  }
  /* the number of scales follows the image size: no pyramid level below ~16 px (main.c:152-157) */
  nscales = nlk_tvl1_scales(nx, ny, nscales, zfactor);
  if (nscales < fscale) fscale = nscales;
  if (verbose)
    fprintf(stderr, "nproc=%d tau=%f lambda=%f theta=%f nscales=%d zfactor=%f nwarps=%d epsilon=%g\n",
            nproc, tau, lambda, theta, nscales, zfactor, nwarps, epsilon);

  nlk_ctx *c = nlkalman_hip_context();
  const size_t n = (size_t)nx * ny;
  void *d_im = NULL, *d_g0 = NULL, *d_g1 = NULL, *d_flow = NULL;
  const size_t cmax = (size_t)(c0 > c1 ? c0 : c1);
  if (cli_dev_alloc(c, &d_im, n * cmax * sizeof(float)) || cli_dev_alloc(c, &d_g0, n * sizeof(float)) ||
      cli_dev_alloc(c, &d_g1, n * sizeof(float)) || cli_dev_alloc(c, &d_flow, 2 * n * sizeof(float)))
    return fail_hip("allocation", c);
  if (nlk_h2d(c, d_im, I0, n * c0 * sizeof(float)) || nlk_dev_gray(c, (float *)d_g0, (float *)d_im, nx, ny, c0) ||
      nlk_sync(c) ||
      nlk_h2d(c, d_im, I1, n * c1 * sizeof(float)) || nlk_dev_gray(c, (float *)d_g1, (float *)d_im, nx, ny, c1))
    return fail_hip("upload", c);
  struct nlk_tvl1_params P = {tau, lambda, theta, nscales, fscale, zfactor, nwarps, epsilon};
  int iters = 0;
  if (nlk_dev_tvl1_flow(c, (float *)d_flow, (float *)d_g0, (float *)d_g1, nx, ny, &P, &iters))
    return fail_hip("flow", c);
  float *flow = (float *)cli_host_keep(malloc(2 * n * sizeof(float)));
  if (!flow || nlk_d2h(c, flow, d_flow, 2 * n * sizeof(float))) return fail_hip("download", c);
  if (verbose) fprintf(stderr, "Iterations: %d\n", iters);
  if (img_write(outfile, flow, nx, ny, 2)) {
    fprintf(stderr, "ERROR: could not write \"%s\"\n", outfile);
    return EXIT_FAILURE;
  }
  /* (device buffers: released with the process, or by the resident server after the request - cli_server.h) */
  return EXIT_SUCCESS;
}

#ifndef NLK_TOOL_NO_MAIN
int main(int argc, const char **argv) {
  const int remote = cli_remote("tvl1flow", argc, argv); /* a resident server (NLK_SERVER) does the work, if there is one */
  return remote >= 0 ? remote : nlk_tool_tvl1(argc, argv);
}
#endif
