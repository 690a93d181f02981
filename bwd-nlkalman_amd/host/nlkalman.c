/* nlkalman.c — host side (plain C) of the drop-in API in include/nlkalman.h.
 *
 * Same six symbols, prototypes and calling conventions as the reference's
 * src/nlkalman.h:14-53: host pointers in, host pointers out, void returns,
 * fatal errors print to stderr and exit(1) (reference: src/nlkalman.c:165-177).
 * All arithmetic on images happens in the HIP kernels behind include/nlk_hip.h;
 * there is NO CPU fallback: without a usable GPU every frame function aborts.
 *
 * A process-wide device context is created on first use (the CLI makes one
 * call per process; a long-running caller reuses the context and its scratch
 * buffers) and torn down at exit. NLK_DEVICE selects the HIP device (default 0);
 * NLK_DEVICES=0,1,... splits the two frame functions over several devices (multidev.c).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nlk_hip.h"
#include "nlkalman.h"
#include "tvl1flow.h"

static nlk_ctx *g_ctx = NULL;
int nlk_multi_first_device(void); /* multidev.c */

static struct { void *p; size_t cap; } g_slot[4]; /* device buffers of the frame calls */

static void ctx_atexit(void) {
  if (g_ctx) {
    for (int i = 0; i < 4; ++i)
      if (g_slot[i].p) nlk_dev_free(g_ctx, g_slot[i].p);
    nlk_ctx_destroy(g_ctx);
  }
  g_ctx = NULL;
}

static void die(const char *what, nlk_ctx *c) {
  fprintf(stderr, "nlkalman (hip): %s: %s\n", what, nlk_last_error(c));
  exit(1);
}

/* The process-wide context is created once, under a lock: the command-line tools bring it up on a second thread while
 * they read their inputs (cli_args.c: cli_warm_start), and the resident server may see a request's first call while
 * that thread is still at it. `may_exit` = 0 (the warm-up thread): a failure is only recorded - the thread that needs
 * the context tries again and is the ONE that prints and exits (two threads in exit() is undefined behaviour). */
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;

static nlk_ctx *ctx_get(int may_exit) {
  pthread_mutex_lock(&g_ctx_lock);
  if (!g_ctx) {
    /* NLK_DEVICE, else the first index of NLK_DEVICES (a list of one device, or a call the list cannot split), else 0 */
    const char *dev = getenv("NLK_DEVICE");
    const int first = nlk_multi_first_device();
    if (nlk_ctx_create(&g_ctx, dev ? atoi(dev) : (first >= 0 ? first : 0)) != NLK_OK) {
      g_ctx = NULL;
      pthread_mutex_unlock(&g_ctx_lock);
      if (may_exit) die("cannot initialise the GPU", NULL);
      return NULL;
    }
    atexit(ctx_atexit);
  }
  nlk_ctx *c = g_ctx;
  pthread_mutex_unlock(&g_ctx_lock);
  return c;
}

static nlk_ctx *ctx(void) { return ctx_get(1); }

nlk_ctx *nlkalman_hip_context(void) { return ctx(); } /* for the CLIs */
int nlkalman_hip_context_warm(void) { return ctx_get(0) ? 0 : -1; } /* the CLIs' warm-up thread: never exits */

/* Device buffers of the image helpers, kept between calls like the frame calls' (hipMalloc / hipFree
 * cost ~0.5 ms per call at 1080p). The API is not re-entrant, like the reference's (its FFTW plans are
 * global, src/nlkalman.c:570-575). */
static float *slot(nlk_ctx *c, int i, size_t bytes) {
  if (g_slot[i].cap < bytes) {
    if (g_slot[i].p) nlk_dev_free(c, g_slot[i].p);
    g_slot[i].p = NULL;
    g_slot[i].cap = 0;
    if (nlk_dev_alloc(c, &g_slot[i].p, bytes)) die("device buffers", c);
    g_slot[i].cap = bytes;
  }
  return (float *)g_slot[i].p;
}

static float *upload(nlk_ctx *c, int i, const float *h, size_t n) {
  if (!h) return NULL;
  float *d = slot(c, i, n * sizeof(float));
  if (nlk_h2d(c, d, h, n * sizeof(float))) die("upload", c);
  return d;
}

/* reference: src/nlkalman.c:92-110 */
void rgb2opp(float *im, int w, int h, int ch) {
  if (ch != 3) return;
  nlk_ctx *c = ctx();
  const size_t n = (size_t)w * h * ch;
  float *d = upload(c, 0, im, n);
  if (nlk_dev_rgb2opp(c, d, w, h, ch) || nlk_d2h(c, im, d, n * sizeof(float))) die("rgb2opp", c);
}

/* reference: src/nlkalman.c:112-130 */
void opp2rgb(float *im, int w, int h, int ch) {
  if (ch != 3) return;
  nlk_ctx *c = ctx();
  const size_t n = (size_t)w * h * ch;
  float *d = upload(c, 0, im, n);
  if (nlk_dev_opp2rgb(c, d, w, h, ch) || nlk_d2h(c, im, d, n * sizeof(float))) die("opp2rgb", c);
}

/* reference: src/nlkalman.c:71-88 */
void warp_bicubic(float *imw, float *im, float *of, float *msk, int w, int h, int ch) {
  nlk_ctx *c = ctx();
  const size_t n = (size_t)w * h;
  float *d_im = upload(c, 0, im, n * ch), *d_of = upload(c, 1, of, n * 2), *d_msk = upload(c, 2, msk, n);
  float *d_out = slot(c, 3, n * ch * sizeof(float));
  if (nlk_dev_warp_bicubic(c, d_out, d_im, d_of, d_msk, w, h, ch) || nlk_d2h(c, imw, d_out, n * ch * sizeof(float)))
    die("warp_bicubic", c);
}

/* reference: src/nlkalman.c:426-487 — sigma-dependent defaults for fields < 0.
 * The expressions keep the reference's int/float/double mix. */
void nlkalman_default_params(struct nlkalman_params *p, float sigma, enum FILTER_MODE mode) {
  if (p->patch_sz < 0) p->patch_sz = 8;
  if (p->search_sz_x < 0) p->search_sz_x = 10;
  if (p->search_sz_t < 0) p->search_sz_t = 5;
  if (p->dista_lambda < 0) p->dista_lambda = 1.0;
  switch (mode) {
    case FLT1:
      if (p->npatches_x < 0) p->npatches_x = (int)(0.5 * sigma + 40.);
      if (p->beta_x < 0) p->beta_x = -0.04 * sigma + 3.91;
      if (p->npatches_t < 0) p->npatches_t = 30;
      if (p->npatches_tagg < 0) p->npatches_tagg = 20;
      if (p->beta_t < 0) p->beta_t = -0.005 * sigma + 2.05;
      break;
    case FLT2:
      if (p->npatches_x < 0) p->npatches_x = (int)(0.5 * sigma + 10.);
      if (p->beta_x < 0) p->beta_x = 0.004 * sigma + 0.21;
      if (p->npatches_t < 0) p->npatches_t = (int)(5 > sigma ? 5 : sigma);
      if (p->npatches_tagg < 0) p->npatches_tagg = 1;
      if (p->beta_t < 0) p->beta_t = 0.014 * sigma + 1.38;
      break;
    case SMO1:
      if (p->npatches_x < 0) p->npatches_x = 0;
      if (p->beta_x < 0) p->beta_x = 0;
      if (p->npatches_t < 0) {
        const float v = 3 * sigma - 15;
        p->npatches_t = (int)(5 > v ? 5 : v);
      }
      if (p->npatches_tagg < 0) p->npatches_tagg = p->npatches_t;
      if (p->beta_t < 0) {
        const double v = -0.14 * sigma + 8.0;
        p->beta_t = 1.0 > v ? 1.0 : v;
      }
      break;
  }
}

/* host/multidev.c: the same call split over the devices of NLK_DEVICES */
int nlk_multi_devices(void);
int nlk_multi_frame(int smoother, float *out, const float *cur, const float *prev, const float *basic, int w, int h,
                    int ch, float sigma, const struct nlkalman_params *P);

static void frame_call(int smoother, float *out, float *cur, float *prev, float *basic, int w,
                       int h, int ch, float sigma, const struct nlkalman_params *prms) {
  if (nlk_multi_devices() > 1 && nlk_multi_frame(smoother, out, cur, prev, basic, w, h, ch, sigma, prms)) return;
  /* host images in, host image out: the frame crosses PCIe in row bands while the bands before it are being
   * matched and filtered (nlk_filter_frame_host, csrc/nlk_hip.hip) */
  nlk_ctx *c = ctx();
  const int rc = smoother ? nlk_smooth_frame_host(c, out, cur, prev, basic, w, h, ch, sigma, prms)
                          : nlk_filter_frame_host(c, out, cur, prev, basic, w, h, ch, sigma, prms);
  if (rc) die(smoother ? "nlkalman_smooth_frame" : "nlkalman_filter_frame", c);
}

/* reference: src/nlkalman.c:518-951 */
void nlkalman_filter_frame(float *deno1, float *nisy1, float *deno0, float *bsic1, int w, int h,
                           int ch, float sigma, const struct nlkalman_params prms, int frame) {
  (void)frame;
  frame_call(0, deno1, nisy1, deno0, bsic1, w, h, ch, sigma, &prms);
}

/* reference: src/nlkalman.c:1409-1865 */
void nlkalman_smooth_frame(float *smoo1, float *filt1, float *smoo0, float *bsic1, int w, int h,
                           int ch, float sigma, const struct nlkalman_params prms, int frame) {
  (void)frame;
  frame_call(1, smoo1, filt1, smoo0, bsic1, w, h, ch, sigma, &prms);
}

/* reference: lib/tvl1flow/tvl1flow_lib.c:345-474 — host pointers in, planar u1 / u2 out */
void Dual_TVL1_optic_flow_multiscale(float *I0, float *I1, float *u1, float *u2, const int nxx,
                                     const int nyy, const float tau, const float lambda,
                                     const float theta, const int nscales, const int fscale,
                                     const float zfactor, const int warps, const float epsilon,
                                     const bool verbose) {
  (void)verbose;
  nlk_ctx *c = ctx();
  const size_t n = (size_t)nxx * nyy;
  struct nlk_tvl1_params P = {tau, lambda, theta, nscales, fscale, zfactor, warps, epsilon};
  float *d0 = upload(c, 0, I0, n), *d1 = upload(c, 1, I1, n);
  float *d_flow = slot(c, 2, 2 * n * sizeof(float));
  float *flow = (float *)malloc(2 * n * sizeof(float));
  if (!flow) die("Dual_TVL1_optic_flow_multiscale", c);
  if (nlk_dev_tvl1_flow(c, d_flow, d0, d1, nxx, nyy, &P, NULL) ||
      nlk_d2h(c, flow, d_flow, 2 * n * sizeof(float)))
    die("Dual_TVL1_optic_flow_multiscale", c);
  for (size_t i = 0; i < n; ++i) {
    u1[i] = flow[2 * i];
    u2[i] = flow[2 * i + 1];
  }
  free(flow);
}
