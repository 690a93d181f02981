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
 * flows and masks are always recomputed (it reuses files left by a previous run).
 *
 * File I/O runs beside the GPU: the output files (float TIFF / .flo / PNG encoding is most of a
 * frame's wall time) are written by a pool of threads from copies of the downloaded arrays, and
 * the next input frame is decoded while the current one is filtered. NLK_SEQ_IO_THREADS sets
 * the pool size (default 6; 0 = write in line). */
#include <errno.h>
#include <pthread.h>
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

/* ---- write-behind: a bounded queue of (path, array) jobs served by worker threads. The arrays
 * are page-locked buffers of one frame each, recycled through a pool: downloads run at the
 * link's rate and nothing is allocated per file */
struct wjob { char *path; float *data; int w, h, ch; struct wjob *next; };
#define POOL_MAX 16
static struct { float *buf[POOL_MAX]; int n; size_t bytes; } P;  /* free buffers (under Q.mu) */
static struct {
  pthread_mutex_t mu;
  pthread_cond_t more, less;
  struct wjob *head, *tail;
  int pending, closed, failed, nthreads;
  pthread_t th[32];
} Q = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, NULL, 0, 0, 0, 0, {0}};
#define WQ_MAX_PENDING 12 /* arrays waiting or being written = pool size - 1 (25 MB each at 1080p RGB) */

static void *wq_worker(void *arg) {
  (void)arg;
  for (;;) {
    pthread_mutex_lock(&Q.mu);
    while (!Q.head && !Q.closed) pthread_cond_wait(&Q.more, &Q.mu);
    struct wjob *j = Q.head;
    if (!j) { pthread_mutex_unlock(&Q.mu); return NULL; }
    Q.head = j->next;
    if (!Q.head) Q.tail = NULL;
    pthread_mutex_unlock(&Q.mu);
    const int bad = img_write(j->path, j->data, j->w, j->h, j->ch);
    if (bad) fprintf(stderr, "nlkalman-seq: cannot write %s\n", j->path);
    float *data = j->data;
    free(j->path);
    free(j);
    pthread_mutex_lock(&Q.mu);
    P.buf[P.n++] = data;  /* back to the pool */
    Q.failed |= bad != 0;
    --Q.pending;
    pthread_cond_signal(&Q.less);
    pthread_mutex_unlock(&Q.mu);
  }
}

static void wq_start(void) {
  const char *e = getenv("NLK_SEQ_IO_THREADS");
  Q.nthreads = e ? atoi(e) : 6;
  if (Q.nthreads > 32) Q.nthreads = 32;
  for (int i = 0; i < Q.nthreads; ++i)
    if (pthread_create(&Q.th[i], NULL, wq_worker, NULL)) { Q.nthreads = i; break; }
}

/* a free frame buffer (waits for a writer to return one) */
static float *pool_get(size_t bytes) {
  if (bytes > P.bytes) { fprintf(stderr, "nlkalman-seq: internal: buffer of %zu bytes asked\n", bytes); exit(1); }
  pthread_mutex_lock(&Q.mu);
  while (P.n == 0) pthread_cond_wait(&Q.less, &Q.mu);
  float *b = P.buf[--P.n];
  pthread_mutex_unlock(&Q.mu);
  return b;
}

static void pool_init(size_t bytes) {
  P.bytes = bytes;
  const int want = Q.nthreads > 0 ? WQ_MAX_PENDING + 1 : 1;
  for (P.n = 0; P.n < want && P.n < POOL_MAX; ++P.n) {
    void *h = NULL;
    CHK(nlk_host_alloc(C, &h, bytes));
    P.buf[P.n] = (float *)h;
  }
}

/* takes ownership of `path` (malloc'ed) and `data` (from the pool) */
static void wq_write(char *path, float *data, int w, int h, int ch) {
  if (Q.nthreads == 0) {
    if (img_write(path, data, w, h, ch)) { fprintf(stderr, "nlkalman-seq: cannot write %s\n", path); exit(1); }
    P.buf[P.n++] = data;
    free(path);
    return;
  }
  struct wjob *j = malloc(sizeof *j);
  j->path = path; j->data = data; j->w = w; j->h = h; j->ch = ch; j->next = NULL;
  pthread_mutex_lock(&Q.mu);
  if (Q.tail) Q.tail->next = j; else Q.head = j;
  Q.tail = j;
  ++Q.pending;
  pthread_cond_signal(&Q.more);
  pthread_mutex_unlock(&Q.mu);
}

/* waits for every file; returns nonzero if one could not be written */
static int wq_finish(void) {
  pthread_mutex_lock(&Q.mu);
  Q.closed = 1;
  pthread_cond_broadcast(&Q.more);
  pthread_mutex_unlock(&Q.mu);
  for (int i = 0; i < Q.nthreads; ++i) pthread_join(Q.th[i], NULL);
  return Q.failed;
}

/* ---- read-ahead: the next input frame is decoded by a helper thread */
static struct { pthread_t th; int active; char name[1024]; float *data; int w, h, ch; float *pinned; size_t pinned_bytes; } R;
static void *ra_worker(void *arg) {
  (void)arg;
  R.data = img_read(R.name, &R.w, &R.h, &R.ch);
  const size_t bytes = (size_t)R.w * R.h * R.ch * sizeof(float);
  if (R.data && R.pinned && bytes == R.pinned_bytes) {  /* stage it where the upload is fast */
    memcpy(R.pinned, R.data, bytes);
    free(R.data);
    R.data = R.pinned;
  }
  return NULL;
}
static void ra_start(const char *name) {
  snprintf(R.name, sizeof R.name, "%s", name);
  R.active = pthread_create(&R.th, NULL, ra_worker, NULL) == 0;
}
static float *ra_get(const char *name, int *w, int *h, int *ch) {
  if (R.active && strcmp(name, R.name) == 0) {
    pthread_join(R.th, NULL);
    R.active = 0;
    *w = R.w; *h = R.h; *ch = R.ch;
    return R.data;
  }
  return img_read(name, w, h, ch);
}

/* download a device array into a pool buffer and queue it for writing */
static void write_dev(char *path, const float *d, int w, int h, int ch) {
  const size_t bytes = (size_t)w * h * ch * sizeof(float);
  float *host = pool_get(bytes);
  CHK(nlk_d2h(C, host, d, bytes));
  wq_write(path, host, w, h, ch);
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

/* RGB copy of an opponent-space device frame -> file (takes ownership of `path`) */
static void write_frame(char *path, const float *d_opp, float *d_tmp, int w, int h, int ch) {
  const size_t bytes = (size_t)w * h * ch * sizeof(float);
  CHK(nlk_d2d(C, d_tmp, d_opp, bytes));
  CHK(nlk_dev_opp2rgb(C, d_tmp, w, h, ch));
  write_dev(path, d_tmp, w, h, ch);
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
  wq_start();
  int w = 0, h = 0, ch = 0;
  size_t bytes = 0;
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
    float *fr = ra_get(name, &w1, &h1, &c1);
    if (!fr) return 1;
    if (t == 0) {
      w = w1; h = h1; ch = c1;
      bytes = (size_t)w * h * ch * sizeof(float);
      d_rgb = dev_frame(bytes); d_noisy = dev_frame(bytes); d_tmp = dev_frame(bytes); d_warp = dev_frame(bytes);
      d_g0 = dev_frame((size_t)w * h * 4); d_g1 = dev_frame((size_t)w * h * 4); d_occ = dev_frame((size_t)w * h * 4);
      d_flow = dev_frame((size_t)w * h * 8);
      pool_init(bytes > (size_t)w * h * 8 ? bytes : (size_t)w * h * 8);
      if (Q.nthreads > 0) {
        void *hp = NULL;
        CHK(nlk_host_alloc(C, &hp, bytes));
        R.pinned = (float *)hp;
        R.pinned_bytes = bytes;
      }
    } else if (w1 != w || h1 != h || c1 != ch) {
      fprintf(stderr, "nlkalman-seq: %s: frame size differs from the first frame\n", name);
      return 1;
    }
    CHK(nlk_h2d(C, d_rgb, fr, bytes));  /* (returns when the copy is done: the staging buffer is free again) */
    if (fr != R.pinned) free(fr);
    if (i + stp <= lfr && Q.nthreads > 0) {  /* decode the next frame meanwhile */
      char next[1024];
      snprintf(next, sizeof next, seq, i + stp);
      ra_start(next);
    }
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
      write_dev(path_of(out, "bflo1-%03d.flo", i), d_flow, w, h, 2);
      write_dev(path_of(out, "bocc1-%03d.png", i), d_occ, w, h, 1);
    }
    write_frame(path_of(out, "flt1-%03d.tif", i), n1, d_tmp, w, h, ch);
    write_frame(path_of(out, "flt2-%03d.tif", i), n2, d_tmp, w, h, ch);
    if (flt1) nlk_dev_free(C, flt1);
    flt1 = n1;
    flt2[t] = n2;
    if (!smoothing && t > 0) { nlk_dev_free(C, flt2[t - 1]); flt2[t - 1] = NULL; }
    if (verbose) printf("frame %d filtered\n", i);
  }
  if (!smoothing) return wq_finish(); /* script line 113 */

  /* ---- backward pass (script lines 117-150) */
  float **smo = calloc(nframes, sizeof(float *));
  smo[nframes - 1] = flt2[nframes - 1];
  write_frame(path_of(out, "smo1-%03d.tif", ffr + (nframes - 1) * stp), smo[nframes - 1], d_tmp, w, h, ch);
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
    write_dev(path_of(out, "fflo-%03d.flo", i), d_flow, w, h, 2);
    write_dev(path_of(out, "focc-%03d.png", i), d_occ, w, h, 1);
    write_frame(path_of(out, "smo1-%03d.tif", i), smo[t], d_tmp, w, h, ch);
    if (verbose) printf("frame %d smoothed\n", i);
  }
  return wq_finish();
}
