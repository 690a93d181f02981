/* main_seq.c — `nlkalman-seq`: the whole recursion of scripts/nlkalman-seq.sh in ONE process
 * with the frames resident on the GPU. Same positional arguments and the
 * same files in the output folder as the script:
 *
 *   nlkalman-seq SEQ FFR LFR SIG OUT [STP [FPM [SPM [OPM]]]]
 *     SEQ   printf pattern of the noisy frames (e.g. in/%03d.tif)      (script: $1)
 *     FFR, LFR, STP  first / last frame, frame step (default 1)         ($2, $3, $6)
 *     SIG   noise standard deviation                                    ($4)
 *     OUT   output folder: flt1-%03d.tif flt2-%03d.tif bflo1-%03d.flo bocc1-%03d.png and,
 *           unless SPM is "no", fflo-%03d.flo focc-%03d.png smo1-%03d.tif  ($5)
 *     FPM   extra nlkalman-flt options (--f1_p ... --f2_l ..., one string)  ($7)
 *     SPM   extra nlkalman-smo options (--s1_p ...), or "no": no smoothing  ($8)
 *     OPM   "FSCALE1 DW1 TH1 FSCALE2 DW2 TH2": finest flow scale, flow data weight (lambda)
 *           and occlusion threshold of the forward / backward pass
 *           (default "1 0.25 0.75 1 0.25 0.75", script line 11)
 *
 * What the script does with four processes and ~10 image files per frame (reference:
 * scripts/nlkalman-seq.sh:30-150) happens here through the device C-ABI: per frame
 * tvl1flow(noisy_t -> flt2_{t-1}) -> occlusion mask -> warp + FLT1 -> warp + FLT2, then
 * backwards tvl1flow(flt2_t -> smo1_{t+1}) -> mask -> warp + SMO1. Frames stay in the
 * opponent colour space between steps (the script's processes convert to RGB files and back:
 * a 1e-5 rounding on the 0..255 scale is the only numerical difference). Unlike the script,
 * flows and masks are always recomputed (it reuses files left by a previous run). */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "cli_args.h"
#include "imgio.h"
#include "nlk_hip.h"
#include "nlkalman.h"

nlk_ctx *nlkalman_hip_context(void); /* libnlkalman.so: the process-wide device context */

static nlk_ctx *C;
#define CHK(call)                                                    \
  do {                                                               \
    if ((call) != NLK_OK) {                                          \
      fprintf(stderr, "nlkalman-seq: %s\n", nlk_last_error(C));      \
      exit(1);                                                       \
    }                                                                \
  } while (0)

static void unset(struct nlkalman_params *p) {
  p->patch_sz = p->search_sz_x = p->search_sz_t = -1;
  p->npatches_x = p->npatches_t = p->npatches_tagg = -1;
  p->dista_lambda = p->beta_x = p->beta_t = -1.f;
}

/* "a b  c" -> argv {prog, a, b, c}; returns argc */
static int split(const char *prog, const char *s, const char ***argv_out) {
  char *buf = strdup(s ? s : "");
  int n = 1, cap = 64;
  const char **av = malloc(sizeof(char *) * cap);
  av[0] = prog;
  for (char *t = strtok(buf, " \t\n"); t; t = strtok(NULL, " \t\n")) {
    if (n + 1 >= cap) av = realloc(av, sizeof(char *) * (cap *= 2));
    av[n++] = t;
  }
  *argv_out = av;
  return n;
}

static float *dev_frame(size_t bytes) {
  void *d = NULL;
  CHK(nlk_dev_alloc(C, &d, bytes));
  return (float *)d;
}

static char *path_of(const char *dir, const char *pattern, int i) {
  char name[256], *full = malloc(strlen(dir) + 300);
  snprintf(name, sizeof name, pattern, i);
  sprintf(full, "%s/%s", dir, name);
  return full;
}

/* RGB copy of an opponent-space device frame -> file */
static void write_frame(const char *path, const float *d_opp, float *d_tmp, float *host, int w, int h, int ch) {
  const size_t bytes = (size_t)w * h * ch * sizeof(float);
  CHK(nlk_d2d(C, d_tmp, d_opp, bytes));
  CHK(nlk_dev_opp2rgb(C, d_tmp, w, h, ch));
  CHK(nlk_d2h(C, host, d_tmp, bytes));
  if (img_write(path, host, w, h, ch)) { fprintf(stderr, "nlkalman-seq: cannot write %s\n", path); exit(1); }
}

int main(int argc, const char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s SEQ FFR LFR SIG OUT [STP [FPM [SPM [OPM]]]]\n"
                    "  one-process equivalent of scripts/nlkalman-seq.sh (see the header of main_seq.c)\n", argv[0]);
    return 1;
  }
  const char *seq = argv[1], *out = argv[5];
  const int ffr = atoi(argv[2]), lfr = atoi(argv[3]);
  const float sigma = atof(argv[4]);
  const int stp = argc > 6 && atoi(argv[6]) > 0 ? atoi(argv[6]) : 1;
  const char *fpm = argc > 7 ? argv[7] : "", *spm = argc > 8 ? argv[8] : "";
  const char *opm = argc > 9 && argv[9][0] ? argv[9] : "1 0.25 0.75 1 0.25 0.75";
  int fs1 = 1, fs2 = 1;
  float dw1 = 0.25f, th1 = 0.75f, dw2 = 0.25f, th2 = 0.75f;
  if (sscanf(opm, "%d %f %f %d %f %f", &fs1, &dw1, &th1, &fs2, &dw2, &th2) != 6) {
    fprintf(stderr, "nlkalman-seq: OPM must hold 6 numbers: FSCALE1 DW1 TH1 FSCALE2 DW2 TH2\n");
    return 1;
  }
  const int smoothing = strcmp(spm, "no") != 0;

  /* filter / smoother parameters: the options of nlkalman-flt and nlkalman-smo, same grammar */
  struct nlkalman_params f1, f2, s1;
  unset(&f1); unset(&f2); unset(&s1);
  int verbose = 0;
  const struct cli_option fopts[] = {
      {CLI_INT, 0, "f1_p", &f1.patch_sz, "patch size"},
      {CLI_INT, 0, "f1_sx", &f1.search_sz_x, "search radius (spatial filtering)"},
      {CLI_INT, 0, "f1_st", &f1.search_sz_t, "search radius (temporal filtering)"},
      {CLI_INT, 0, "f1_nx", &f1.npatches_x, "number of similar patches spatial"},
      {CLI_INT, 0, "f1_nt", &f1.npatches_t, "number of similar patches kalman"},
      {CLI_INT, 0, "f1_nt_agg", &f1.npatches_tagg, "number of similar patches kalman spatial average"},
      {CLI_FLOAT, 0, "f1_bx", &f1.beta_x, "noise multiplier in spatial filtering"},
      {CLI_FLOAT, 0, "f1_bt", &f1.beta_t, "noise multiplier in kalman filtering"},
      {CLI_FLOAT, 0, "f1_l", &f1.dista_lambda, "noisy patch weight in patch distance"},
      {CLI_INT, 0, "f2_p", &f2.patch_sz, "patch size"},
      {CLI_INT, 0, "f2_sx", &f2.search_sz_x, "search radius (spatial filtering)"},
      {CLI_INT, 0, "f2_st", &f2.search_sz_t, "search radius (temporal filtering)"},
      {CLI_INT, 0, "f2_nx", &f2.npatches_x, "number of similar patches spatial"},
      {CLI_INT, 0, "f2_nt", &f2.npatches_t, "number of similar patches kalman"},
      {CLI_INT, 0, "f2_nt_agg", &f2.npatches_tagg, "number of similar patches kalman spatial average"},
      {CLI_FLOAT, 0, "f2_bx", &f2.beta_x, "noise multiplier in spatial filtering"},
      {CLI_FLOAT, 0, "f2_bt", &f2.beta_t, "noise multiplier in kalman filtering"},
      {CLI_FLOAT, 0, "f2_l", &f2.dista_lambda, "noisy patch weight in patch distance"},
      {CLI_INT, 'v', "verbose", &verbose, "verbose output"},
      {CLI_END, 0, NULL, NULL, NULL}};
  const struct cli_option sopts[] = {
      {CLI_INT, 0, "s1_p", &s1.patch_sz, "patch size"},
      {CLI_INT, 0, "s1_st", &s1.search_sz_t, "search region radius"},
      {CLI_INT, 0, "s1_nt", &s1.npatches_t, "number of similar patches kalman"},
      {CLI_INT, 0, "s1_nt_agg", &s1.npatches_tagg, "number of similar patches kalman spatial average"},
      {CLI_FLOAT, 0, "s1_bt", &s1.beta_t, "noise multiplier in kalman filtering"},
      {CLI_FLOAT, 0, "s1_l", &s1.dista_lambda, "noisy patch weight in patch distance"},
      {CLI_INT, 'v', "verbose", &verbose, "verbose output"},
      {CLI_END, 0, NULL, NULL, NULL}};
  const char **av;
  int ac = split("nlkalman-seq (FPM)", fpm, &av);
  cli_parse(fopts, "nlkalman-seq (FPM)", "filtering parameters", ac, av);
  if (smoothing) {
    ac = split("nlkalman-seq (SPM)", spm, &av);
    cli_parse(sopts, "nlkalman-seq (SPM)", "smoothing parameters", ac, av);
  }
  if (f1.patch_sz == 0 || f2.patch_sz == 0) {
    fprintf(stderr, "nlkalman-seq: both filtering iterations are needed (f1_p, f2_p != 0)\n");
    return 1;
  }
  nlkalman_default_params(&f1, sigma, FLT1);
  nlkalman_default_params(&f2, sigma, FLT2);
  nlkalman_default_params(&s1, sigma, SMO1);

  /* every input frame must exist (script lines 19-28) */
  int nframes = 0;
  for (int i = ffr; i <= lfr; i += stp, ++nframes) {
    char name[1024];
    snprintf(name, sizeof name, seq, i);
    FILE *f = fopen(name, "rb");
    if (!f) { printf("ERROR: %s not found\n", name); return 1; }
    fclose(f);
  }
  if (nframes < 1) { fprintf(stderr, "nlkalman-seq: empty frame range\n"); return 1; }
  if (mkdir(out, 0777) && errno != EEXIST) { perror(out); return 1; }

  C = nlkalman_hip_context();
  int w = 0, h = 0, ch = 0;
  size_t bytes = 0;
  float *host = NULL, *host2 = NULL;
  float *d_rgb = NULL, *d_noisy = NULL, *d_tmp = NULL, *d_warp = NULL, *d_g0 = NULL, *d_g1 = NULL;
  float *d_flow = NULL, *d_occ = NULL, *flt1 = NULL;
  float **flt2 = calloc(nframes, sizeof(float *));  /* kept for the backward pass */
  struct nlk_tvl1_params of;

  /* ---- forward pass (script lines 30-115) */
  int t = 0;
  for (int i = ffr; i <= lfr; i += stp, ++t) {
    char name[1024];
    snprintf(name, sizeof name, seq, i);
    int w1, h1, c1;
    float *fr = img_read(name, &w1, &h1, &c1);
    if (!fr) return 1;
    if (t == 0) {
      w = w1; h = h1; ch = c1;
      bytes = (size_t)w * h * ch * sizeof(float);
      host = malloc(bytes);
      host2 = malloc((size_t)w * h * 2 * sizeof(float));
      d_rgb = dev_frame(bytes); d_noisy = dev_frame(bytes); d_tmp = dev_frame(bytes); d_warp = dev_frame(bytes);
      d_g0 = dev_frame((size_t)w * h * 4); d_g1 = dev_frame((size_t)w * h * 4); d_occ = dev_frame((size_t)w * h * 4);
      d_flow = dev_frame((size_t)w * h * 8);
    } else if (w1 != w || h1 != h || c1 != ch) {
      fprintf(stderr, "nlkalman-seq: %s: frame size differs from the first frame\n", name);
      return 1;
    }
    CHK(nlk_h2d(C, d_rgb, fr, bytes));
    free(fr);
    CHK(nlk_d2d(C, d_noisy, d_rgb, bytes));
    CHK(nlk_dev_rgb2opp(C, d_noisy, w, h, ch));
    float *n1 = dev_frame(bytes), *n2 = dev_frame(bytes);
    if (t == 0) {
      CHK(nlk_dev_filter_frame(C, n1, d_noisy, NULL, NULL, w, h, ch, sigma, &f1));
      CHK(nlk_dev_filter_frame(C, n2, d_noisy, NULL, n1, w, h, ch, sigma, &f2));
    } else {
      /* backward flow noisy_t -> flt2_{t-1}, occlusion mask (script lines 57-73) */
      nlk_tvl1_default_params(&of);
      of.lambda = dw1; of.fscale = fs1;
      of.nscales = nlk_tvl1_scales(w, h, of.nscales, of.zfactor);
      if (of.nscales < of.fscale) of.fscale = of.nscales;
      CHK(nlk_dev_gray(C, d_g0, d_rgb, w, h, ch));
      CHK(nlk_d2d(C, d_tmp, flt2[t - 1], bytes));
      CHK(nlk_dev_opp2rgb(C, d_tmp, w, h, ch));
      CHK(nlk_dev_gray(C, d_g1, d_tmp, w, h, ch));
      CHK(nlk_dev_tvl1_flow(C, d_flow, d_g0, d_g1, w, h, &of, NULL));
      CHK(nlk_dev_occlusion_mask(C, d_occ, d_flow, w, h, th1));
      CHK(nlk_dev_warp_bicubic(C, d_warp, flt1, d_flow, d_occ, w, h, ch));
      CHK(nlk_dev_filter_frame(C, n1, d_noisy, d_warp, NULL, w, h, ch, sigma, &f1));
      CHK(nlk_dev_warp_bicubic(C, d_warp, flt2[t - 1], d_flow, d_occ, w, h, ch));
      CHK(nlk_dev_filter_frame(C, n2, d_noisy, d_warp, n1, w, h, ch, sigma, &f2));
      char *p = path_of(out, "bflo1-%03d.flo", i);
      CHK(nlk_d2h(C, host2, d_flow, (size_t)w * h * 8));
      if (img_write(p, host2, w, h, 2)) return fprintf(stderr, "cannot write %s\n", p), 1;
      free(p);
      p = path_of(out, "bocc1-%03d.png", i);
      CHK(nlk_d2h(C, host2, d_occ, (size_t)w * h * 4));
      if (img_write(p, host2, w, h, 1)) return fprintf(stderr, "cannot write %s\n", p), 1;
      free(p);
    }
    char *p = path_of(out, "flt1-%03d.tif", i);
    write_frame(p, n1, d_tmp, host, w, h, ch);
    free(p);
    p = path_of(out, "flt2-%03d.tif", i);
    write_frame(p, n2, d_tmp, host, w, h, ch);
    free(p);
    if (flt1) nlk_dev_free(C, flt1);
    flt1 = n1;
    flt2[t] = n2;
    if (!smoothing && t > 0) { nlk_dev_free(C, flt2[t - 1]); flt2[t - 1] = NULL; }
    if (verbose) printf("frame %d filtered\n", i);
  }
  if (!smoothing) return 0; /* script line 113 */

  /* ---- backward pass (script lines 117-150) */
  float **smo = calloc(nframes, sizeof(float *));
  smo[nframes - 1] = flt2[nframes - 1];
  {
    char *p = path_of(out, "smo1-%03d.tif", ffr + (nframes - 1) * stp);
    write_frame(p, smo[nframes - 1], d_tmp, host, w, h, ch);
    free(p);
  }
  for (t = nframes - 2; t >= 0; --t) {
    const int i = ffr + t * stp;
    nlk_tvl1_default_params(&of);
    of.lambda = dw2; of.fscale = fs2;
    of.nscales = nlk_tvl1_scales(w, h, of.nscales, of.zfactor);
    if (of.nscales < of.fscale) of.fscale = of.nscales;
    CHK(nlk_d2d(C, d_rgb, flt2[t], bytes));
    CHK(nlk_dev_opp2rgb(C, d_rgb, w, h, ch));
    CHK(nlk_dev_gray(C, d_g0, d_rgb, w, h, ch));
    CHK(nlk_d2d(C, d_tmp, smo[t + 1], bytes));
    CHK(nlk_dev_opp2rgb(C, d_tmp, w, h, ch));
    CHK(nlk_dev_gray(C, d_g1, d_tmp, w, h, ch));
    CHK(nlk_dev_tvl1_flow(C, d_flow, d_g0, d_g1, w, h, &of, NULL));
    CHK(nlk_dev_occlusion_mask(C, d_occ, d_flow, w, h, th2));
    CHK(nlk_dev_warp_bicubic(C, d_warp, smo[t + 1], d_flow, d_occ, w, h, ch));
    smo[t] = dev_frame(bytes);
    CHK(nlk_dev_smooth_frame(C, smo[t], flt2[t], d_warp, NULL, w, h, ch, sigma, &s1));
    char *p = path_of(out, "fflo-%03d.flo", i);
    CHK(nlk_d2h(C, host2, d_flow, (size_t)w * h * 8));
    if (img_write(p, host2, w, h, 2)) return fprintf(stderr, "cannot write %s\n", p), 1;
    free(p);
    p = path_of(out, "focc-%03d.png", i);
    CHK(nlk_d2h(C, host2, d_occ, (size_t)w * h * 4));
    if (img_write(p, host2, w, h, 1)) return fprintf(stderr, "cannot write %s\n", p), 1;
    free(p);
    p = path_of(out, "smo1-%03d.tif", i);
    write_frame(p, smo[t], d_tmp, host, w, h, ch);
    free(p);
    if (verbose) printf("frame %d smoothed\n", i);
  }
  return 0;
}
