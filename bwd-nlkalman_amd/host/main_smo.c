/* nlkalman-smo — backward NL-Kalman (RTS-like) smoothing of one frame.
 *
 * Same command line and messages as the reference tool (reference:
 * src/main-smo.c:53-79 options, :87-96 checks, :98-124 verbose dump, :130-190
 * inputs, :193-213 run), so scripts/nlkalman-seq.sh:147-149 runs unchanged.
 * Exit status: the reference returns 1 on SUCCESS (src/main-smo.c:222) and its
 * pipelines ignore the status; this tool returns 0 on success — set
 * NLK_SMO_REFERENCE_EXIT=1 to get the reference's value. The frames stay in
 * HBM across colour transform, warp and smoothing (include/nlk_hip.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cli_args.h"
#include "cli_server.h"
#include "imgio.h"
#include "nlk_hip.h"
#include "nlkalman.h"

nlk_ctx *nlkalman_hip_context(void);

#define CHK(call)                                                              \
  do {                                                                         \
    if ((call) != NLK_OK) {                                                    \
      fprintf(stderr, "nlkalman-smo: %s\n", nlk_last_error(c));                \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static float *to_dev(nlk_ctx *c, const float *h, size_t n) {
  void *d = NULL;
  if (!h) return NULL;
  if (cli_dev_alloc(c, &d, n * sizeof(float)) || nlk_h2d(c, d, h, n * sizeof(float))) {
    fprintf(stderr, "nlkalman-smo: %s\n", nlk_last_error(c));
    cli_exit(1);
  }
  return (float *)d;
}

static int tool_body(int argc, const char **argv);

/* the tool as a function: main() below, or the resident server (main_server.c). Whichever way the body returns -
 * there are a dozen early `return 1` in it, as in the reference's main - the warm-up thread is joined (nobody reaches
 * exit() while it is inside hipInit) and the host images are released. */
int nlk_tool_smo(int argc, const char **argv) {
  const int rc = tool_body(argc, argv);
  cli_warm_join();
  cli_host_release();
  return rc;
}

static int tool_body(int argc, const char **argv) {
  const char *flt1_path = NULL, *smo0_path = NULL, *fflo_path = NULL, *focc_path = NULL, *smo1_path = NULL;
  float sigma = 0.f;
  int verbose = 0;
  struct nlkalman_params s1;
  s1.patch_sz = s1.search_sz_x = s1.search_sz_t = -1;
  s1.npatches_x = s1.npatches_t = s1.npatches_tagg = -1;
  s1.dista_lambda = s1.beta_x = s1.beta_t = -1.f;

  const struct cli_option options[] = {
      {CLI_GROUP, 0, NULL, NULL, "Data i/o options"},
      {CLI_STRING, 0, "flt1", &flt1_path, "input filtered frame path"},
      {CLI_STRING, 0, "smo0", &smo0_path, "input next smoothed frame path"},
      {CLI_STRING, 'o', "fflo", &fflo_path, "input fwd flow path"},
      {CLI_STRING, 'k', "focc", &focc_path, "input fwd occlusion mask path"},
      {CLI_STRING, 0, "smo1", &smo1_path, "output smoothed frame"},
      {CLI_FLOAT, 's', "sigma", &sigma, "noise standard dev"},
      {CLI_GROUP, 0, NULL, NULL, "Smoothing options"},
      {CLI_INT, 0, "s1_p", &s1.patch_sz, "patch size"},
      {CLI_INT, 0, "s1_st", &s1.search_sz_t, "search region radius"},
      {CLI_INT, 0, "s1_nt", &s1.npatches_t, "number of similar patches kalman"},
      {CLI_INT, 0, "s1_nt_agg", &s1.npatches_tagg, "number of similar patches kalman spatial average"},
      {CLI_FLOAT, 0, "s1_bt", &s1.beta_t, "noise multiplier in kalman filtering"},
      {CLI_FLOAT, 0, "s1_l", &s1.dista_lambda, "noisy patch weight in patch distance"},
      {CLI_GROUP, 0, NULL, NULL, "Program options"},
      {CLI_INT, 'v', "verbose", &verbose, "verbose output"},
      {CLI_END, 0, NULL, NULL, NULL}};
  cli_parse(options, "nlkalman-smo", "Patch-based Kalman smoother for video denoising.", argc, argv);

  if (!smo1_path) return fprintf(stderr, "Error: no output path given\n"), 1;
  if (s1.patch_sz == 0) return fprintf(stderr, "Error: s1_p == 0\n"), 1;
  nlkalman_default_params(&s1, sigma, SMO1);
  cli_warm_start(); /* the GPU comes up while the input files are read */

  if (verbose)
    printf("data input:\n\tnoise         %05.2f\n\tfiltering 1   %s\n\tfiltering 0   %s\n"
           "\tfwd flows     %s\n\tfwd occlus.   %s\n\ndata output:\n\tsmoothing 1   %s\n\n"
           "smoother params:\n\tpatch      %d\n\tsearch_t   %d\n\tnp_t       %d\n\tnp_tagg    %d\n"
           "\tlambda     %g\n\tbeta_t     %g\n\n",
           sigma, flt1_path, smo0_path, fflo_path, focc_path, smo1_path, s1.patch_sz,
           s1.search_sz_t, s1.npatches_t, s1.npatches_tagg, s1.dista_lambda, s1.beta_t);

  int w, h, ch, w1, h1, c1;
  float *flt1 = (float *)cli_host_keep(img_read(flt1_path, &w, &h, &ch));
  if (!flt1) return fprintf(stderr, "Opening %s failed\n", flt1_path), 1;
  float *smo0 = (float *)cli_host_keep(img_read(smo0_path, &w1, &h1, &c1));
  if (!smo0) return fprintf(stderr, "Opening %s failed\n", smo0_path), 1;
  if (w * h * ch != w1 * h1 * c1) return fprintf(stderr, "Filtered frames size missmatch\n"), 1;
  float *fflo = NULL, *focc = NULL;
  if (fflo_path) {
    fflo = (float *)cli_host_keep(img_read(fflo_path, &w1, &h1, &c1));
    if (!fflo) return fprintf(stderr, "Opening %s failed\n", fflo_path), 1;
    if (w * h != w1 * h1 || c1 != 2) return fprintf(stderr, "Frame and optical flow size missmatch\n"), 1;
  }
  if (fflo_path && focc_path) {
    focc = (float *)cli_host_keep(img_read(focc_path, &w1, &h1, &c1));
    if (!focc) return fprintf(stderr, "Opening %s failed\n", focc_path), 1;
    if (w * h != w1 * h1 || c1 != 1) return fprintf(stderr, "Frame and occlusion mask size missmatch\n"), 1;
  }

  cli_warm_join();
  nlk_ctx *c = nlkalman_hip_context();
  const size_t n = (size_t)w * h * ch, bytes = n * sizeof(float);
  float *d_flt1 = to_dev(c, flt1, n), *d_smo0 = to_dev(c, smo0, n);
  float *d_flo = to_dev(c, fflo, (size_t)w * h * 2), *d_occ = to_dev(c, focc, (size_t)w * h);
  void *d_warp = NULL, *d_smo1 = NULL;
  CHK(cli_dev_alloc(c, &d_warp, bytes));
  CHK(cli_dev_alloc(c, &d_smo1, bytes));
  CHK(nlk_dev_rgb2opp(c, d_flt1, w, h, ch));
  CHK(nlk_dev_rgb2opp(c, d_smo0, w, h, ch));
  const float *prev = d_smo0;
  if (d_flo) {
    CHK(nlk_dev_warp_bicubic(c, (float *)d_warp, d_smo0, d_flo, d_occ, w, h, ch));
    prev = (const float *)d_warp;
  }
  CHK(nlk_dev_smooth_frame(c, (float *)d_smo1, d_flt1, prev, NULL, w, h, ch, sigma, &s1));
  CHK(nlk_dev_opp2rgb(c, (float *)d_smo1, w, h, ch));
  float *host = (float *)cli_host_keep(malloc(bytes));
  CHK(nlk_d2h(c, host, d_smo1, bytes));
  if (img_write(smo1_path, host, w, h, ch)) return fprintf(stderr, "cannot write %s\n", smo1_path), 1;
  const char *e = getenv("NLK_SMO_REFERENCE_EXIT");
  return cli_leave((e && e[0] == '1') ? 1 : 0);
}

#ifndef NLK_TOOL_NO_MAIN
int main(int argc, const char **argv) {
  const int remote = cli_remote("nlkalman-smo", argc, argv); /* a resident server (NLK_SERVER) does the work, if there is one */
  return remote >= 0 ? remote : nlk_tool_smo(argc, argv);
}
#endif
