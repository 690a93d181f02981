/* multidev.c — the frame functions of the drop-in API split over several GPUs, in C.
 *
 * NLK_DEVICES=0,1,2,... (a comma-separated list of HIP device indices; an index may repeat, which is
 * how the one-GPU test box exercises this path) makes nlkalman_filter_frame / nlkalman_smooth_frame
 * cut the frame into row strips of the patch grid, one per listed device, exactly like the
 * torch.distributed driver (bwd-nlkalman_amd/strips.py) does over RCCL, but from one host thread
 * that feeds the devices' streams:
 *   1. every device gets its strip + search halo of the caller's host images (the previous-frame
 *      halo needs no exchange here: the whole previous frame is in host memory);
 *   2. nlk_dev_strip_match on every strip -> one 64-bit mark word per target; the words of all
 *      strips are collected into one host array and handed to every device, which replays the
 *      raster-order mask over the WHOLE grid (nlk_dev_mask_commit): decisions identical to the
 *      single-GPU / serial order for any number of devices;
 *   3. nlk_dev_strip_group with the strip's slice of the decisions;
 *   4. accumulator rows written outside a strip's own rows go to the neighbour that owns them
 *      (hipMemcpyPeerAsync between the two devices) and are added there;
 *   5. every device normalises and returns its own rows.
 * Reference analogue: the static row split of the OpenMP loop, src/nlkalman.c:586. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
static int g_ndev = -1; /* -1: NLK_DEVICES not looked at yet */

static void md_die(const char *what, nlk_ctx *c) {
  fprintf(stderr, "nlkalman (hip, %d devices): %s: %s\n", g_ndev, what, nlk_last_error(c));
  exit(1);
}

static void md_atexit(void) {
  for (int d = 0; d < g_ndev; ++d) {
    buf_t *b[] = {&g_dev[d].cur, &g_dev[d].prev, &g_dev[d].basic, &g_dev[d].out, &g_dev[d].acc, &g_dev[d].marks,
                  &g_dev[d].marks_full, &g_dev[d].active, &g_dev[d].rtop, &g_dev[d].rbot};
    for (unsigned i = 0; i < sizeof b / sizeof b[0]; ++i)
      if (b[i]->p) nlk_dev_free(g_dev[d].c, b[i]->p);
    nlk_ctx_destroy(g_dev[d].c);
  }
  g_ndev = 0;
}

/* number of devices listed in NLK_DEVICES (0 or 1: the single-device path of nlkalman.c) */
int nlk_multi_devices(void) {
  if (g_ndev >= 0) return g_ndev;
  g_ndev = 0;
  const char *s = getenv("NLK_DEVICES");
  if (!s || !*s) return 0;
  int ids[NLK_MAXDEV], n = 0;
  while (*s && n < NLK_MAXDEV) {
    char *e;
    const long v = strtol(s, &e, 10);
    if (e == s) break;
    ids[n++] = (int)v;
    s = *e == ',' ? e + 1 : e;
  }
  if (n < 2) return 0;
  for (int d = 0; d < n; ++d) {
    memset(&g_dev[d], 0, sizeof g_dev[d]);
    if (nlk_ctx_create(&g_dev[d].c, ids[d]) != NLK_OK) {
      g_ndev = d;
      md_die("cannot initialise a device of NLK_DEVICES", NULL);
    }
  }
  g_ndev = n;
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

void nlk_multi_frame(int smoother, float *out, const float *cur, const float *prev, const float *basic, int w, int h,
                     int ch, float sigma, const struct nlkalman_params *P) {
  const int psz = P->patch_sz, step = psz / 2;
  if (psz < 2 || w < psz || h < psz) { fprintf(stderr, "nlkalman (hip): bad patch size / image size\n"); exit(1); }
  const int ngx = (w - psz) / step + 1, ngy = (h - psz) / step + 1;
  const int halo = smoother ? P->search_sz_t : imax(P->search_sz_x, P->search_sz_t);
  int n = g_ndev;
  while (n > 1 && ngy / n < 1) --n;
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
  for (int d = 0; d + 1 < n; ++d)
    if (g_dev[d].Y1 > g_dev[d + 1].own1 || g_dev[d + 1].Y0 < g_dev[d].own0) {
      fprintf(stderr, "nlkalman (hip): strips thinner than the search halo: list fewer devices in NLK_DEVICES\n");
      exit(1);
    }
  const size_t row = (size_t)w * ch * sizeof(float), nmark = (size_t)ngx * ngy;
  unsigned long long *marks_host = (unsigned long long *)malloc(nmark * sizeof *marks_host);
  if (!marks_host) { fprintf(stderr, "nlkalman (hip): out of memory\n"); exit(1); }
  int reach = 0;
  /* 1 + 2a: upload, match */
  for (int d = 0; d < n; ++d) {
    mdev_t *D = &g_dev[d];
    const int hl = D->Y1 - D->Y0;
    float *dc = (float *)grow(D, &D->cur, row * hl);
    float *dp = prev ? (float *)grow(D, &D->prev, row * hl) : NULL;
    float *db = basic ? (float *)grow(D, &D->basic, row * hl) : NULL;
    grow(D, &D->out, row * hl);
    grow(D, &D->acc, (size_t)(ch + 1) * hl * w * sizeof(float));
    grow(D, &D->marks, (size_t)(D->gy1 - D->gy0) * ngx * 8);
    grow(D, &D->marks_full, nmark * 8);
    grow(D, &D->active, nmark);
    if (nlk_h2d(D->c, dc, cur + (size_t)D->Y0 * w * ch, row * hl) ||
        (prev && nlk_h2d(D->c, dp, prev + (size_t)D->Y0 * w * ch, row * hl)) ||
        (basic && nlk_h2d(D->c, db, basic + (size_t)D->Y0 * w * ch, row * hl)))
      md_die("upload", D->c);
    if (nlk_dev_strip_match(D->c, dc, dp, db, w, hl, ch, sigma, P, D->gy0 * step - D->Y0, D->gy1 - D->gy0, smoother,
                            D->marks.p, &reach))
      md_die("strip_match", D->c);
  }
  /* 2b: the mark words of every strip -> every device; whole-grid replay on each */
  for (int d = 0; d < n; ++d)
    if (nlk_d2h(g_dev[d].c, marks_host + (size_t)g_dev[d].gy0 * ngx, g_dev[d].marks.p,
                (size_t)(g_dev[d].gy1 - g_dev[d].gy0) * ngx * 8))
      md_die("mark words", g_dev[d].c);
  for (int d = 0; d < n; ++d) {
    mdev_t *D = &g_dev[d];
    const int hl = D->Y1 - D->Y0;
    if (nlk_h2d(D->c, D->marks_full.p, marks_host, nmark * 8) ||
        nlk_dev_mask_commit(D->c, D->marks_full.p, ngx, ngy, reach, (unsigned char *)D->active.p) ||
        nlk_dev_zero(D->c, D->acc.p, (size_t)(ch + 1) * hl * w * sizeof(float)) ||
        nlk_dev_strip_group(D->c, (float *)D->acc.p, (unsigned char *)D->active.p + (size_t)D->gy0 * ngx))
      md_die("mask_commit / strip_group", D->c);
  }
  free(marks_host);
  /* 4: accumulator rows outside the own rows -> the neighbour that owns them (device to device) */
  for (int d = 0; d < n; ++d) {
    mdev_t *D = &g_dev[d];
    const int hl = D->Y1 - D->Y0;
    if (d > 0) { /* rows [Y0, own0) belong to device d-1 */
      mdev_t *U = &g_dev[d - 1];
      const int nr = D->own0 - D->Y0, hu = U->Y1 - U->Y0;
      float *rb = (float *)grow(U, &U->rbot, (size_t)(ch + 1) * nr * w * sizeof(float));
      for (int p = 0; p <= ch; ++p)
        if (nlk_dev_copy_peer(U->c, rb + (size_t)p * nr * w, D->c, (float *)D->acc.p + (size_t)p * hl * w,
                              (size_t)nr * w * sizeof(float)))
          md_die("halo copy", U->c);
      (void)hu;
    }
    if (d + 1 < n) { /* rows [own1, Y1) belong to device d+1 */
      mdev_t *L = &g_dev[d + 1];
      const int nr = D->Y1 - D->own1;
      float *rt = (float *)grow(L, &L->rtop, (size_t)(ch + 1) * nr * w * sizeof(float));
      for (int p = 0; p <= ch; ++p)
        if (nlk_dev_copy_peer(L->c, rt + (size_t)p * nr * w, D->c,
                              (float *)D->acc.p + ((size_t)p * hl + (D->own1 - D->Y0)) * w, (size_t)nr * w * sizeof(float)))
          md_die("halo copy", L->c);
    }
  }
  for (int d = 0; d < n; ++d) {
    mdev_t *D = &g_dev[d];
    const int hl = D->Y1 - D->Y0;
    if (d > 0) { /* what device d-1 wrote into my first rows: its rows [own1(d-1), Y1(d-1)) = my [own0, ...) */
      const int nr = g_dev[d - 1].Y1 - g_dev[d - 1].own1;
      for (int p = 0; p <= ch; ++p)
        if (nlk_dev_add(D->c, (float *)D->acc.p + ((size_t)p * hl + (D->own0 - D->Y0)) * w,
                        (float *)D->rtop.p + (size_t)p * nr * w, (size_t)nr * w))
          md_die("halo add", D->c);
    }
    if (d + 1 < n) { /* what device d+1 wrote above its own rows: its rows [Y0(d+1), own0(d+1)) = my last rows */
      const int nr = g_dev[d + 1].own0 - g_dev[d + 1].Y0;
      for (int p = 0; p <= ch; ++p)
        if (nlk_dev_add(D->c, (float *)D->acc.p + ((size_t)p * hl + (g_dev[d + 1].Y0 - D->Y0)) * w,
                        (float *)D->rbot.p + (size_t)p * nr * w, (size_t)nr * w))
          md_die("halo add", D->c);
    }
    /* 5: own rows */
    if (nlk_dev_frame_normalize(D->c, (float *)D->out.p, (float *)D->acc.p, (float *)D->cur.p, w, hl, ch,
                                D->own0 - D->Y0, D->own1 - D->Y0))
      md_die("normalize", D->c);
  }
  for (int d = 0; d < n; ++d) {
    mdev_t *D = &g_dev[d];
    if (nlk_d2h(D->c, out + (size_t)D->own0 * w * ch, (float *)D->out.p + (size_t)(D->own0 - D->Y0) * w * ch,
                row * (D->own1 - D->own0)))
      md_die("download", D->c);
  }
}
