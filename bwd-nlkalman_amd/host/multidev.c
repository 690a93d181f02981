/* multidev.c — the frame functions of the drop-in API split over several GPUs, in C.
 *
 * NLK_DEVICES=0,1,2,... (a comma-separated list of HIP device indices; an index may repeat, which is
 * how the one-GPU test box exercises this path) makes nlkalman_filter_frame / nlkalman_smooth_frame
 * cut the frame into row strips of the patch grid, one per listed device, exactly like the
 * torch.distributed driver (bwd-nlkalman_amd/strips.py) does over RCCL, with one host thread per
 * device (the devices upload, match, filter and download side by side; three barriers per call):
 *   1. every device gets its strip + search halo of the caller's host images (the previous-frame
 *      halo needs no exchange here: the whole previous frame is in host memory);
 *   2. nlk_dev_strip_match on every strip -> one 64-bit mark word per target; every device pulls the
 *      words of all strips from their devices (peer copies, no host round trip) and replays the
 *      raster-order mask over the WHOLE grid (nlk_dev_mask_commit): decisions identical to the
 *      single-GPU / serial order for any number of devices;
 *   3. nlk_dev_strip_group with the strip's slice of the decisions;
 *   4. accumulator rows written outside a strip's own rows go to the neighbour that owns them
 *      (hipMemcpyPeerAsync between the two devices) and are added there;
 *   5. every device normalises and returns its own rows.
 * Reference analogue: the static row split of the OpenMP loop, src/nlkalman.c:586. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "nlk_hip.h"
#include "nlkalman.h"

#define NLK_MAXDEV 16

typedef struct { void *p; size_t cap; } buf_t;
typedef struct {
  nlk_ctx *c;
  buf_t cur, prev, basic, out, acc, marks, marks_full, active, rtop, rbot;
  int gy0, gy1, Y0, Y1, own0, own1;
} mdev_t;

static mdev_t g_dev[NLK_MAXDEV];
static int g_ids[NLK_MAXDEV];
static int g_ndev = -1; /* -1: NLK_DEVICES not looked at yet */
static int g_nlisted = 0;

/* One host thread per device (the listed devices work side by side: uploads over their own PCIe links,
 * kernels, peer copies); the caller's thread is device 0's. Helpers are started once and sleep between
 * frame calls. */
typedef struct {
  int smoother, w, h, ch, n;
  float sigma;
  const struct nlkalman_params *P;
  float *out;
  const float *cur, *prev, *basic;
} job_t;
static job_t g_job;
static pthread_t g_thr[NLK_MAXDEV];
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER, g_peer_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static pthread_barrier_t g_bar;
static unsigned g_gen = 0;
static int g_quit = 0, g_reach[NLK_MAXDEV], g_fail_at = -1;

/* Fatal error on any device thread. exit() would run md_atexit, which joins the workers - but they are
 * parked in pthread_barrier_wait (nothing releases a barrier short of every party arriving), the caller
 * may BE a worker (joining itself), and two threads in exit() at once is undefined. So: message, flush,
 * and leave without the exit handlers; the driver reclaims the device memory of a dead process. */
static void md_die(const char *what, nlk_ctx *c) {
  fprintf(stderr, "nlkalman (hip, %d devices): %s: %s\n", g_ndev, what, nlk_last_error(c));
  fflush(stderr);
  fflush(stdout);
  _exit(1);
}

static void md_atexit(void) {
  if (g_ndev > 1) {
    pthread_mutex_lock(&g_mu);
    g_quit = 1;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
    for (int d = 1; d < g_ndev; ++d) pthread_join(g_thr[d], NULL);
  }
  for (int d = 0; d < g_ndev; ++d) {
    buf_t *b[] = {&g_dev[d].cur, &g_dev[d].prev, &g_dev[d].basic, &g_dev[d].out, &g_dev[d].acc, &g_dev[d].marks,
                  &g_dev[d].marks_full, &g_dev[d].active, &g_dev[d].rtop, &g_dev[d].rbot};
    for (unsigned i = 0; i < sizeof b / sizeof b[0]; ++i)
      if (b[i]->p) nlk_dev_free(g_dev[d].c, b[i]->p);
    nlk_ctx_destroy(g_dev[d].c);
  }
  g_ndev = 0;
}

static void run_device(int d);
static void *worker(void *arg) {
  const int d = (int)(long)arg;
  unsigned seen = 0;
  for (;;) {
    pthread_mutex_lock(&g_mu);
    while (g_gen == seen && !g_quit) pthread_cond_wait(&g_cv, &g_mu);
    const int quit = g_quit;
    seen = g_gen;
    pthread_mutex_unlock(&g_mu);
    if (quit) return NULL;
    run_device(d);
  }
}

static void parse_list(void) {
  if (g_nlisted) return;
  const char *s = getenv("NLK_DEVICES");
  if (!s) return;
  while (*s && g_nlisted < NLK_MAXDEV) {
    char *e;
    const long v = strtol(s, &e, 10);
    if (e == s) break;
    g_ids[g_nlisted++] = (int)v;
    s = *e == ',' ? e + 1 : e;
  }
}

/* first index listed in NLK_DEVICES, or -1: the device of the single-device path when NLK_DEVICE is not set */
int nlk_multi_first_device(void) {
  parse_list();
  return g_nlisted > 0 ? g_ids[0] : -1;
}

/* number of devices listed in NLK_DEVICES (0 or 1: the single-device path of nlkalman.c) */
int nlk_multi_devices(void) {
  if (g_ndev >= 0) return g_ndev;
  g_ndev = 0;
  parse_list();
  const int n = g_nlisted;
  if (n < 2) return 0;
  for (int d = 0; d < n; ++d) {
    memset(&g_dev[d], 0, sizeof g_dev[d]);
    if (nlk_ctx_create(&g_dev[d].c, g_ids[d]) != NLK_OK) {
      g_ndev = d;
      md_die("cannot initialise a device of NLK_DEVICES", NULL);
    }
  }
  g_ndev = n;
  if (getenv("NLK_MULTI_TEST_FAIL")) g_fail_at = atoi(getenv("NLK_MULTI_TEST_FAIL"));
  pthread_barrier_init(&g_bar, NULL, (unsigned)n);
  for (int d = 1; d < n; ++d)
    if (pthread_create(&g_thr[d], NULL, worker, (void *)(long)d)) md_die("cannot start a device thread", NULL);
  atexit(md_atexit);
  return n;
}

static void *grow(mdev_t *D, buf_t *b, size_t bytes) {
  if (b->cap < bytes) {
    if (b->p) nlk_dev_free(D->c, b->p);
    b->p = NULL;
    b->cap = 0;
    if (nlk_dev_alloc(D->c, &b->p, bytes)) md_die("device buffers", D->c);
    b->cap = bytes;
  }
  return b->p;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* device-to-device copy ordered behind everything enqueued on the source's stream (the event it
 * records lives in the SOURCE context and several devices may pull from one source: one at a time) */
static void pull(mdev_t *D, void *dst, mdev_t *S, const void *src, size_t n, const char *what) {
  pthread_mutex_lock(&g_peer_mu);
  const int rc = nlk_dev_copy_peer(D->c, dst, S->c, src, n);
  pthread_mutex_unlock(&g_peer_mu);
  if (rc) md_die(what, D->c);
}

/* the strip of device d, start to end; the devices meet at three barriers (strips matched / groups
 * filtered / neighbours' halo rows taken) */
static void run_device(int d) {
  const job_t *J = &g_job;
  const int n = J->n, w = J->w, h = J->h, ch = J->ch, smoother = J->smoother;
  const struct nlkalman_params *P = J->P;
  const int psz = P->patch_sz, step = psz / 2;
  const int ngx = (w - psz) / step + 1, ngy = (h - psz) / step + 1;
  const size_t row = (size_t)w * ch * sizeof(float), nmark = (size_t)ngx * ngy;
  mdev_t *D = &g_dev[d];
  const int on = d < n;
  const int hl = on ? D->Y1 - D->Y0 : 0;
  (void)h;
  /* 1 + 2a: upload, match */
  if (on) {
    float *dc = (float *)grow(D, &D->cur, row * hl);
    float *dp = J->prev ? (float *)grow(D, &D->prev, row * hl) : NULL;
    float *db = J->basic ? (float *)grow(D, &D->basic, row * hl) : NULL;
    grow(D, &D->out, row * hl);
    grow(D, &D->acc, (size_t)(ch + 1) * hl * w * sizeof(float));
    grow(D, &D->marks, (size_t)(D->gy1 - D->gy0) * ngx * 8);
    grow(D, &D->marks_full, nmark * 8);
    grow(D, &D->active, nmark);
    if (d > 0) grow(D, &D->rtop, (size_t)(ch + 1) * (g_dev[d - 1].Y1 - g_dev[d - 1].own1) * w * sizeof(float));
    if (d + 1 < n) grow(D, &D->rbot, (size_t)(ch + 1) * (g_dev[d + 1].own0 - g_dev[d + 1].Y0) * w * sizeof(float));
    if (nlk_h2d(D->c, dc, J->cur + (size_t)D->Y0 * w * ch, row * hl) ||
        (J->prev && nlk_h2d(D->c, dp, J->prev + (size_t)D->Y0 * w * ch, row * hl)) ||
        (J->basic && nlk_h2d(D->c, db, J->basic + (size_t)D->Y0 * w * ch, row * hl)))
      md_die("upload", D->c);
    if (nlk_dev_strip_match(D->c, dc, dp, db, w, hl, ch, J->sigma, P, D->gy0 * step - D->Y0, D->gy1 - D->gy0, smoother,
                            D->marks.p, &g_reach[d]))
      md_die("strip_match", D->c);
  }
  pthread_barrier_wait(&g_bar);
  /* (test hook, tests/test_gpu_parity.py: NLK_MULTI_TEST_FAIL=<d> makes device d report an error here - with
   * the other device threads on their way to the next barrier - to show that the process leaves with status 1
   * instead of hanging) */
  if (g_fail_at == d) md_die("injected failure (NLK_MULTI_TEST_FAIL)", D->c);
  /* 2b: the mark words of every strip, device to device; whole-grid replay on each; 3: the strip's groups */
  if (on) {
    for (int e = 0; e < n; ++e)
      pull(D, (unsigned long long *)D->marks_full.p + (size_t)g_dev[e].gy0 * ngx, &g_dev[e], g_dev[e].marks.p,
           (size_t)(g_dev[e].gy1 - g_dev[e].gy0) * ngx * 8, "mark words");
    if (nlk_dev_mask_commit(D->c, D->marks_full.p, ngx, ngy, g_reach[d], (unsigned char *)D->active.p) ||
        nlk_dev_zero(D->c, D->acc.p, (size_t)(ch + 1) * hl * w * sizeof(float)) ||
        nlk_dev_strip_group(D->c, (float *)D->acc.p, (unsigned char *)D->active.p + (size_t)D->gy0 * ngx))
      md_die("mask_commit / strip_group", D->c);
  }
  pthread_barrier_wait(&g_bar);
  /* 4: the accumulator rows the neighbours wrote inside my own rows (device to device), added here */
  if (on) {
    if (d > 0) { /* rows [own1(d-1), Y1(d-1)) of device d-1 = my first own rows */
      mdev_t *U = &g_dev[d - 1];
      const int nr = U->Y1 - U->own1, hu = U->Y1 - U->Y0;
      for (int p = 0; p <= ch; ++p)
        pull(D, (float *)D->rtop.p + (size_t)p * nr * w, U, (float *)U->acc.p + ((size_t)p * hu + (U->own1 - U->Y0)) * w,
             (size_t)nr * w * sizeof(float), "halo copy");
    }
    if (d + 1 < n) { /* rows [Y0(d+1), own0(d+1)) of device d+1 = my last own rows */
      mdev_t *L = &g_dev[d + 1];
      const int nr = L->own0 - L->Y0, hb = L->Y1 - L->Y0;
      for (int p = 0; p <= ch; ++p)
        pull(D, (float *)D->rbot.p + (size_t)p * nr * w, L, (float *)L->acc.p + (size_t)p * hb * w,
             (size_t)nr * w * sizeof(float), "halo copy");
    }
    if (d > 0) {
      const int nr = g_dev[d - 1].Y1 - g_dev[d - 1].own1;
      for (int p = 0; p <= ch; ++p)
        if (nlk_dev_add(D->c, (float *)D->acc.p + ((size_t)p * hl + (D->own0 - D->Y0)) * w,
                        (float *)D->rtop.p + (size_t)p * nr * w, (size_t)nr * w))
          md_die("halo add", D->c);
    }
    if (d + 1 < n) {
      const int nr = g_dev[d + 1].own0 - g_dev[d + 1].Y0;
      for (int p = 0; p <= ch; ++p)
        if (nlk_dev_add(D->c, (float *)D->acc.p + ((size_t)p * hl + (g_dev[d + 1].Y0 - D->Y0)) * w,
                        (float *)D->rbot.p + (size_t)p * nr * w, (size_t)nr * w))
          md_die("halo add", D->c);
    }
    /* 5: own rows */
    if (nlk_dev_frame_normalize(D->c, (float *)D->out.p, (float *)D->acc.p, (float *)D->cur.p, w, hl, ch,
                                D->own0 - D->Y0, D->own1 - D->Y0) ||
        nlk_d2h(D->c, J->out + (size_t)D->own0 * w * ch, (float *)D->out.p + (size_t)(D->own0 - D->Y0) * w * ch,
                row * (D->own1 - D->own0)))
      md_die("normalize / download", D->c);
  }
  /* (nobody reuses a buffer a neighbour may still be pulling from before every device is done) */
  pthread_barrier_wait(&g_bar);
}

/* returns 1 when the call was split over the devices, 0 when this configuration cannot be (a group
 * that reaches more than 3 grid cells: 64-bit mark words cannot describe it; strips thinner than the
 * search halo): the caller then runs it on one device, like any call without NLK_DEVICES */
int nlk_multi_frame(int smoother, float *out, const float *cur, const float *prev, const float *basic, int w, int h,
                    int ch, float sigma, const struct nlkalman_params *P) {
  static int told = 0;
  const int psz = P->patch_sz, step = psz / 2;
  if (psz < 2 || w < psz || h < psz) return 0; /* (the single-device path reports it) */
  const int ngy = (h - psz) / step + 1;
  const int halo = smoother ? P->search_sz_t : imax(P->search_sz_x, P->search_sz_t);
  const int wmark = (smoother || prev) ? P->search_sz_t : P->search_sz_x;
  int n = g_ndev;
  for (; n > 1; --n) {
    if (ngy / n < 1) continue;
    /* strips (the plan of strips.py): rows of the patch grid, pixel rows incl. the search halo, own rows */
    for (int d = 0; d < n; ++d) {
      mdev_t *D = &g_dev[d];
      D->gy0 = (int)((long)ngy * d / n);
      D->gy1 = (int)((long)ngy * (d + 1) / n);
      D->Y0 = imax(0, D->gy0 * step - halo);
      D->Y1 = imin(h, (D->gy1 - 1) * step + halo + psz);
      D->own0 = d > 0 ? D->gy0 * step : 0;
      D->own1 = d < n - 1 ? D->gy1 * step : h;
    }
    int thin = 0; /* a strip may only spill into its direct neighbour's own rows */
    for (int d = 0; d + 1 < n; ++d)
      if (g_dev[d].Y1 > g_dev[d + 1].own1 || g_dev[d + 1].Y0 < g_dev[d].own0) thin = 1;
    if (!thin) break;
  }
  if (n < 2 || wmark / step > 3) {
    if (!told)
      fprintf(stderr, "nlkalman (hip): NLK_DEVICES: this call is not split (%s); it runs on one device\n",
              n < 2 ? "strips thinner than the search halo" : "group reach above 3 grid cells");
    told = 1;
    return 0;
  }
  g_job = (job_t){smoother, w, h, ch, n, sigma, P, out, cur, prev, basic};
  pthread_mutex_lock(&g_mu);
  ++g_gen;
  pthread_cond_broadcast(&g_cv);
  pthread_mutex_unlock(&g_mu);
  run_device(0);
  return 1;
}
