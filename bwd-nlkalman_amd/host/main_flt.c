/* nlkalman-flt — per-frame NL-Kalman filtering, first and/or second iteration.
 *
 * Same command line, mode logic, messages and exit codes as the reference tool
 * (reference: src/main-flt.c:71-117 options, :129-149 modes, :152-212 verbose
 * dump, :216-332 inputs, :335-388 run + outputs), so the shell pipelines
 * (scripts/nlkalman-seq.sh:39-41,80-81,100-102) run unchanged. What differs is
 * where the work happens: every frame is uploaded once, colour transform,
 * warps and both filtering iterations run on the GPU through the C-ABI of
 * include/nlk_hip.h with the intermediate frames resident in HBM, and only
 * the requested outputs come back.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cli_args.h"
#include "cli_server.h"
#include "imgio.h"
#include "nlk_hip.h"
#include "nlkalman.h"

nlk_ctx *nlkalman_hip_context(void); /* process-wide context of libnlkalman.so */

static void unset(struct nlkalman_params *p) {
  p->patch_sz = p->search_sz_x = p->search_sz_t = -1;
  p->npatches_x = p->npatches_t = p->npatches_tagg = -1;
  p->dista_lambda = p->beta_x = p->beta_t = -1.f;
}

static void dump(const char *title, const struct nlkalman_params *p) {
  printf("%s\n\tpatch      %d\n\tsearch_x   %d\n\tsearch_t   %d\n\tnp_x       %d\n"
         "\tnp_t       %d\n\tnp_tagg    %d\n\tlambda     %g\n\tbeta_x     %g\n\tbeta_t     %g\n\n",
         title, p->patch_sz, p->search_sz_x, p->search_sz_t, p->npatches_x, p->npatches_t,
         p->npatches_tagg, p->dista_lambda, p->beta_x, p->beta_t);
}

#define CHK(call)                                                              \
  do {                                                                         \
    if ((call) != NLK_OK) {                                                    \
      fprintf(stderr, "nlkalman-flt: %s\n", nlk_last_error(c));                \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static float *to_dev(nlk_ctx *c, const float *h, size_t n) {
  void *d = NULL;
  if (!h) return NULL;
  if (cli_dev_alloc(c, &d, n * sizeof(float)) || nlk_h2d(c, d, h, n * sizeof(float))) {
    fprintf(stderr, "nlkalman-flt: %s\n", nlk_last_error(c));
    cli_exit(1);
  }
  return (float *)d;
}

static int tool_body(int argc, const char **argv);

/* the tool as a function: main() below, or the resident server (main_server.c). Whichever way the body returns -
 * there are a dozen early `return 1` in it, as in the reference's main - the warm-up thread is joined (nobody reaches
 * exit() while it is inside hipInit) and the host images are released. */
int nlk_tool_flt(int argc, const char **argv) {
  const int rc = tool_body(argc, argv);
  cli_warm_join();
  cli_host_release();
  return rc;
}

static int tool_body(int argc, const char **argv) {
  const char *noisy_path = NULL, *bflow_path = NULL, *boccl_path = NULL;
  const char *flt10_path = NULL, *flt20_path = NULL, *flt11_path = NULL, *flt21_path = NULL;
  float sigma = 0.f;
  int verbose = 0;
  struct nlkalman_params f1, f2;
  unset(&f1);
  unset(&f2);

  const struct cli_option options[] = {
      {CLI_GROUP, 0, NULL, NULL, "Data i/o options"},
      {CLI_STRING, 'i', "nisy", &noisy_path, "input noisy frames path"},
      {CLI_STRING, 'o', "bflo", &bflow_path, "input bwd flow path"},
      {CLI_STRING, 'k', "bocc", &boccl_path, "input bwd occlusion masks path"},
      {CLI_STRING, 0, "flt10", &flt10_path, "input previous first filtering path"},
      {CLI_STRING, 0, "flt20", &flt20_path, "input previous second filtering path"},
      {CLI_STRING, 0, "flt11", &flt11_path, "input/output first filtering path"},
      {CLI_STRING, 0, "flt21", &flt21_path, "output second filtering path"},
      {CLI_FLOAT, 's', "sigma", &sigma, "noise standard dev"},
      {CLI_GROUP, 0, NULL, NULL, "First filtering options"},
      {CLI_INT, 0, "f1_p", &f1.patch_sz, "patch size"},
      {CLI_INT, 0, "f1_sx", &f1.search_sz_x, "search radius (spatial filtering)"},
      {CLI_INT, 0, "f1_st", &f1.search_sz_t, "search radius (temporal filtering)"},
      {CLI_INT, 0, "f1_nx", &f1.npatches_x, "number of similar patches spatial"},
      {CLI_INT, 0, "f1_nt", &f1.npatches_t, "number of similar patches kalman"},
      {CLI_INT, 0, "f1_nt_agg", &f1.npatches_tagg, "number of similar patches kalman spatial average"},
      {CLI_FLOAT, 0, "f1_bx", &f1.beta_x, "noise multiplier in spatial filtering"},
      {CLI_FLOAT, 0, "f1_bt", &f1.beta_t, "noise multiplier in kalman filtering"},
      {CLI_FLOAT, 0, "f1_l", &f1.dista_lambda, "noisy patch weight in patch distance"},
      {CLI_GROUP, 0, NULL, NULL, "Second filtering options"},
      {CLI_INT, 0, "f2_p", &f2.patch_sz, "patch size"},
      {CLI_INT, 0, "f2_sx", &f2.search_sz_x, "search radius (spatial filtering)"},
      {CLI_INT, 0, "f2_st", &f2.search_sz_t, "search radius (temporal filtering)"},
      {CLI_INT, 0, "f2_nx", &f2.npatches_x, "number of similar patches spatial"},
      {CLI_INT, 0, "f2_nt", &f2.npatches_t, "number of similar patches kalman"},
      {CLI_INT, 0, "f2_nt_agg", &f2.npatches_tagg, "number of similar patches kalman spatial average"},
      {CLI_FLOAT, 0, "f2_bx", &f2.beta_x, "noise multiplier in spatial filtering"},
      {CLI_FLOAT, 0, "f2_bt", &f2.beta_t, "noise multiplier in kalman filtering"},
      {CLI_FLOAT, 0, "f2_l", &f2.dista_lambda, "noisy patch weight in patch distance"},
      {CLI_GROUP, 0, NULL, NULL, "Program options"},
      {CLI_INT, 'v', "verbose", &verbose, "verbose output"},
      {CLI_END, 0, NULL, NULL, NULL}};
  cli_trace("start");
  cli_parse(options, "nlkalman-flt", "Patch-based Kalman filter for video denoising.", argc, argv);

  /* mode (reference: src/main-flt.c:129-149) */
  const int apply_filt1 = f1.patch_sz != 0;
  const int apply_filt2 = f2.patch_sz != 0 && flt21_path;
  if (!apply_filt1 && !apply_filt2) return fprintf(stderr, "Error: nothing to do, exiting\n"), 1;
  if (!apply_filt1 && !flt11_path)
    return fprintf(stderr, "Error: f1_p == 0 and no input path given, exiting\n"), 1;
  if (!flt11_path && !apply_filt2)
    return fprintf(stderr, "Error: no output path given for any computed output - exiting\n"), 1;
  if (!flt11_path && !flt21_path)
    return fprintf(stderr, "Error: s1_p == 0 and no output paths given for filt1 and filt2\n"), 1;
  if (f2.patch_sz == 0 && flt21_path)
    fprintf(stderr, "Warning: f2_p == 0 - no output files will be stored in %s\n", flt21_path);

  nlkalman_default_params(&f1, sigma, FLT1);
  nlkalman_default_params(&f2, sigma, FLT2);
  cli_warm_start(); /* the GPU comes up while the input files are read */

  if (verbose) {
    printf("data input:\n\tnoise         %05.2f\n\tnoisy frames  %s\n\tbwd flows     %s\n"
           "\tbwd occlus.   %s\n\tprev filt 1   %s\n\tprev filt 2   %s\n",
           sigma, noisy_path, bflow_path, boccl_path, flt10_path, flt20_path);
    if (!apply_filt1) printf("\tfiltering 1   %s\n", flt11_path);
    printf("\ndata output:\n");
    if (apply_filt1) printf("\tfiltering 1   %s\n", flt11_path);
    printf("\tfiltering 2   %s\n\n", flt21_path);
    if (apply_filt1) dump("first filtering parameters:", &f1);
    if (apply_filt2) dump("second filtering parameters:", &f2);
  }

  /* inputs (reference: src/main-flt.c:216-332; same messages) */
  int w, h, ch, w1, h1, c1;
  float *nisy = (float *)cli_host_keep(img_read(noisy_path, &w, &h, &ch));
  if (!nisy) return fprintf(stderr, "Error while openning bwd optical flow\n"), 1;
  float *bflo = NULL, *bocc = NULL, *flt10 = NULL, *flt20 = NULL, *flt11 = NULL;
  if (bflow_path) {
    bflo = (float *)cli_host_keep(img_read(bflow_path, &w1, &h1, &c1));
    if (!bflo) return fprintf(stderr, "Error while openning bwd optical flow\n"), 1;
    if (w * h != w1 * h1 || c1 != 2) return fprintf(stderr, "Frame and optical flow size missmatch\n"), 1;
  }
  if (bflow_path && boccl_path) {
    bocc = (float *)cli_host_keep(img_read(boccl_path, &w1, &h1, &c1));
    if (!bocc) return fprintf(stderr, "Error while openning occlusion mask\n"), 1;
    if (w * h != w1 * h1 || c1 != 1) return fprintf(stderr, "Frame and occlusion mask size missmatch\n"), 1;
  }
  if (flt10_path) {
    flt10 = (float *)cli_host_keep(img_read(flt10_path, &w1, &h1, &c1));
    if (!flt10) fprintf(stderr, "Error while openning previous filter 1 output\n");
    if (flt10 && w * h * ch != w1 * h1 * c1)
      return fprintf(stderr, "Frame and previous filter 1 output size missmatch\n"), 1;
  }
  if (flt20_path) {
    flt20 = (float *)cli_host_keep(img_read(flt20_path, &w1, &h1, &c1));
    if (!flt20) fprintf(stderr, "Error while openning previous filter 2 output\n");
    if (flt20 && w * h * ch != w1 * h1 * c1)
      return fprintf(stderr, "Frame and previous filter 2 output size missmatch\n"), 1;
  }
  if (!apply_filt1) {
    flt11 = (float *)cli_host_keep(img_read(flt11_path, &w1, &h1, &c1));
    if (!flt11) return fprintf(stderr, "Error while openning filter 1 output\n"), 1;
    if (w * h * ch != w1 * h1 * c1) return fprintf(stderr, "Frame and filter 1 output size missmatch\n"), 1;
  }

  /* run on the GPU, frames resident (reference: src/main-flt.c:335-388) */
  cli_trace("inputs read");
  cli_warm_join();
  nlk_ctx *c = nlkalman_hip_context();
  cli_trace("device context ready");
  const size_t n = (size_t)w * h * ch, bytes = n * sizeof(float);
  float *d_nisy = to_dev(c, nisy, n), *d_flo = to_dev(c, bflo, (size_t)w * h * 2);
  float *d_occ = to_dev(c, bocc, (size_t)w * h);
  float *d_f10 = to_dev(c, flt10, n), *d_f20 = to_dev(c, flt20, n), *d_f11 = to_dev(c, flt11, n);
  void *d_warp = NULL, *d_f21 = NULL, *tmp = NULL;
  CHK(cli_dev_alloc(c, &d_warp, bytes));
  CHK(nlk_dev_rgb2opp(c, d_nisy, w, h, ch));
  if (d_f10) CHK(nlk_dev_rgb2opp(c, d_f10, w, h, ch));
  if (d_f20) CHK(nlk_dev_rgb2opp(c, d_f20, w, h, ch));

  if (apply_filt1) {
    const float *prev = d_f10;
    if (d_f10 && d_flo) {
      CHK(nlk_dev_warp_bicubic(c, (float *)d_warp, d_f10, d_flo, d_occ, w, h, ch));
      prev = (const float *)d_warp;
    }
    CHK(cli_dev_alloc(c, &tmp, bytes));
    d_f11 = (float *)tmp;
    CHK(nlk_dev_filter_frame(c, d_f11, d_nisy, prev, NULL, w, h, ch, sigma, &f1));
  } else {
    CHK(nlk_dev_rgb2opp(c, d_f11, w, h, ch));
  }

  cli_trace("first iteration enqueued");
  float *host = (float *)cli_host_keep(malloc(bytes));
  if (apply_filt2) {
    const float *prev = d_f20;
    if (d_flo && d_f20) { /* d_warp is free again: FLT1 has consumed it in stream order */
      CHK(nlk_dev_warp_bicubic(c, (float *)d_warp, d_f20, d_flo, d_occ, w, h, ch));
      prev = (const float *)d_warp;
    }
    CHK(cli_dev_alloc(c, &d_f21, bytes));
    CHK(nlk_dev_filter_frame(c, (float *)d_f21, d_nisy, prev, d_f11, w, h, ch, sigma, &f2));
    if (flt11_path) { /* sic: the reference guards this write with flt11_path (:376) */
      CHK(nlk_dev_opp2rgb(c, (float *)d_f21, w, h, ch));
      CHK(nlk_d2h(c, host, d_f21, bytes));
      if (img_write(flt21_path, host, w, h, ch)) return fprintf(stderr, "cannot write %s\n", flt21_path), 1;
    }
  }
  if (apply_filt1 && flt11_path) {
    CHK(nlk_dev_opp2rgb(c, d_f11, w, h, ch));
    CHK(nlk_d2h(c, host, d_f11, bytes));
    if (img_write(flt11_path, host, w, h, ch)) return fprintf(stderr, "cannot write %s\n", flt11_path), 1;
  }
  cli_trace("outputs written");
  return cli_leave(EXIT_SUCCESS);
}

#ifndef NLK_TOOL_NO_MAIN
int main(int argc, const char **argv) {
  const int remote = cli_remote("nlkalman-flt", argc, argv); /* a resident server (NLK_SERVER) does the work, if there is one */
  return remote >= 0 ? remote : nlk_tool_flt(argc, argv);
}
#endif
