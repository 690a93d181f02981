/* nlk_oracle.c — CPU restatement of the per-frame NL-Kalman hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (bwd-nlkalman_amd/,
 * include/) may include, link or call this file. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline ("port").
 *
 * PARITY UNPINNED. The reference (pariasm/bwd-nlkalman) ships no tests, golden
 * vectors or known-answer fixtures for this path and it
 * cannot be built in this image: src/nlkalman.c:6 includes <fftw3.h> and links
 * FFTW3 single precision (un-vendored, no pinned version: CMakeLists.txt:27-28),
 * which is absent here; building it would need a hand-written stand-in for the
 * missing library, which is not a reference build. The arithmetic below is
 * therefore anchored on (i) the reference's source text, cited per function,
 * (ii) FFTW's published definition of REDFT10 / REDFT01 (FFTW manual, "1d Real-even
 * DFTs (DCTs)": REDFT10 Y_k = 2 sum_j X_j cos(pi (j+1/2) k / n); REDFT01
 * Y_k = X_0 + 2 sum_{j>=1} X_j cos(pi j (k+1/2) / n)) combined with the reference's
 * own scaling at its call sites (src/nlkalman.c:204-220, 281-298, 335-353), which
 * together give the orthonormal DCT-II / DCT-III, (iii) known answers
 * derivable from the source (window values, default-parameter table, warp NaN ring)
 * checked in tests/test_oracle.py, and (iv) an independent numpy/scipy restatement
 * (tests/ref_numpy.py, scipy.fft.dctn/idctn norm='ortho') compared on seeded inputs.
 *
 * Every function cites the reference file:line it follows. Structure, names and
 * data layout are this repo's own: the oracle works on the HWC images of the
 * API but keeps planar scratch, an explicit candidate list and a ranked
 * selection instead of the reference's VLAs + qsort.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "nlk_oracle.h"

/* ------------------------------------------------------------------ helpers */

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
/* the reference's max() macro returns its 2nd argument when the 1st is NaN
 * (reference: src/nlkalman.c:15-18) */
static inline float fmax_ref(float a, float b) { return a > b ? a : b; }

/* reference: src/nlkalman.c:426-487 — defaults for every field < 0 */
void nlko_default_params(nlko_params *p, float sigma, int mode) {
  if (p->patch_sz < 0) p->patch_sz = 8;
  if (p->search_sz_x < 0) p->search_sz_x = 10;
  if (p->search_sz_t < 0) p->search_sz_t = 5;
  if (p->dista_lambda < 0) p->dista_lambda = 1.0f;
  if (mode == NLKO_FLT1) {
    if (p->npatches_x < 0) p->npatches_x = (int)(0.5 * sigma + 40.);
    if (p->beta_x < 0) p->beta_x = -0.04 * sigma + 3.91;
    if (p->npatches_t < 0) p->npatches_t = 30;
    if (p->npatches_tagg < 0) p->npatches_tagg = 20;
    if (p->beta_t < 0) p->beta_t = -0.005 * sigma + 2.05;
  } else if (mode == NLKO_FLT2) {
    if (p->npatches_x < 0) p->npatches_x = (int)(0.5 * sigma + 10.);
    if (p->beta_x < 0) p->beta_x = 0.004 * sigma + 0.21;
    if (p->npatches_t < 0) p->npatches_t = (int)(5 > sigma ? 5 : sigma);
    if (p->npatches_tagg < 0) p->npatches_tagg = 1;
    if (p->beta_t < 0) p->beta_t = 0.014 * sigma + 1.38;
  } else if (mode == NLKO_SMO1) {
    if (p->npatches_x < 0) p->npatches_x = 0;
    if (p->beta_x < 0) p->beta_x = 0;
    if (p->npatches_t < 0) {
      float v = 3 * sigma - 15;
      p->npatches_t = (int)(5 > v ? 5 : v);
    }
    if (p->npatches_tagg < 0) p->npatches_tagg = p->npatches_t;
    if (p->beta_t < 0) {
      double v = -0.14 * sigma + 8.0;
      p->beta_t = 1.0 > v ? 1.0 : v;
    }
  }
}

/* reference: src/nlkalman.c:365-419, "gaussian" branch :401-407 —
 * separable window w1[n] = exp(-x^2/2), x = (n - (N-1)/2)/((N-1)/2)/0.4,
 * float intermediates as in the reference (N, N2, s, x are float there) */
void nlko_window(float *W, int psz) {
  float w1[256];
  const float N2 = ((float)psz - 1.) / 2.;
  for (int n = 0; n < psz; ++n) {
    const float s = .4;
    const float x = ((float)n - N2) / N2 / s;
    w1[n] = exp(-.5 * x * x);
  }
  for (int i = 0; i < psz; ++i)
    for (int j = 0; j < psz; ++j) W[i * psz + j] = w1[i] * w1[j];
}

/* reference: src/nlkalman.c:92-110 */
void nlko_rgb2opp(float *im, int w, int h, int ch) {
  if (ch != 3) return;
  const float a = 1.f / sqrtf(3.f), b = 1.f / sqrtf(2.f);
  const float c = 2.f * a * sqrtf(2.f);
  for (long k = 0; k < (long)w * h; ++k) {
    float *p = im + 3 * k;
    const float r = p[0], g = p[1], bl = p[2];
    p[0] = a * (r + g + bl);
    p[1] = b * (r - bl);
    p[2] = c * (0.25f * r - 0.5f * g + 0.25f * bl);
  }
}

/* reference: src/nlkalman.c:112-130 */
void nlko_opp2rgb(float *im, int w, int h, int ch) {
  if (ch != 3) return;
  const float a = 1.f / sqrtf(3.f), b = 1.f / sqrtf(2.f);
  const float c = a / b;
  for (long k = 0; k < (long)w * h; ++k) {
    float *p = im + 3 * k;
    const float y = p[0], u = p[1], v = p[2];
    p[0] = a * y + b * u + 0.5f * c * v;
    p[1] = a * y - c * v;
    p[2] = a * y - b * u + 0.5f * c * v;
  }
}

/* reference: src/nlkalman.c:36-41 — Keys cubic, double arithmetic on float taps */
static float cubic1d(const float v[4], float x) {
  return v[1] + 0.5 * x * (v[2] - v[0] +
                           x * (2.0 * v[0] - 5.0 * v[1] + 4.0 * v[2] - v[3] +
                                x * (3.0 * (v[1] - v[2]) + v[3] - v[0])));
}

/* reference: src/nlkalman.c:29-33,43-88 — backward bicubic warp; a tap outside
 * the image reads NaN, an occluded pixel (msk != 0) is NaN in every channel */
void nlko_warp_bicubic(float *imw, const float *im, const float *of,
                       const float *msk, int w, int h, int ch) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float *o = imw + ((long)x + (long)y * w) * ch;
      if (msk && msk[x + (long)y * w] != 0) {
        for (int c = 0; c < ch; ++c) o[c] = NAN;
        continue;
      }
      float xw = x + of[((long)x + (long)y * w) * 2 + 0];
      float yw = y + of[((long)x + (long)y * w) * 2 + 1];
      xw -= 1;
      yw -= 1;
      const int ix = (int)floor(xw), iy = (int)floor(yw);
      for (int c = 0; c < ch; ++c) {
        float col[4][4]; /* col[i][j]: tap at (ix+i, iy+j) */
        for (int j = 0; j < 4; ++j)
          for (int i = 0; i < 4; ++i) {
            const int sx = ix + i, sy = iy + j;
            col[i][j] = (sx < 0 || sx >= w || sy < 0 || sy >= h)
                            ? NAN
                            : im[((long)sx + (long)sy * w) * ch + c];
          }
        float v[4];
        for (int i = 0; i < 4; ++i) v[i] = cubic1d(col[i], yw - iy);
        o[c] = cubic1d(v, xw - ix);
      }
    }
}

/* reference: lib/imscript-lite/src/random.c:19-31,50-53,68-75 and awgn.c:24-26
 * — Knuth LCG + Box-Muller AWGN, used to build the synthetic benchmark frames */
void nlko_awgn(float *x, long n, float sigma, uint32_t seed) {
  uint64_t s = seed;
  for (long i = 0; i < n; ++i) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const double u1 = (uint32_t)(s >> 32) / (0.0 + UINT32_MAX);
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const double u2 = (uint32_t)(s >> 32) / (0.0 + UINT32_MAX);
    x[i] += sigma * (sqrt(-2 * log(u1)) * cos(2 * M_PI * u2));
  }
}

/* ---------------------------------------------------------------------- DCT */

/* Orthonormal DCT-II basis C[k][j] = s_k cos(pi (j + 1/2) k / n), s_0 = sqrt(1/n),
 * s_k = sqrt(2/n): FFTW REDFT10 (factor 2 per dimension, and 2 for the
 * length-1 third dimension) times the reference's scaling 1/sqrt(8 w h) and
 * 1/sqrt(2) on the first row, first column and the whole (single) t-plane
 * (reference: src/nlkalman.c:204-212, 281-298). */
void nlko_dct_basis(float *C, int n) {
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < n; ++j) {
      const double s = (k == 0) ? sqrt(1.0 / n) : sqrt(2.0 / n);
      C[k * n + j] = (float)(s * cos(M_PI * (j + 0.5) * k / n));
    }
}

/* forward 2-D DCT of one n x n plane: Y = C X C^T (rows, then columns) */
static void dct2_forward(const float *C, int n, float *X, float *tmp) {
  for (int y = 0; y < n; ++y)
    for (int v = 0; v < n; ++v) {
      float acc = 0.f;
      for (int x = 0; x < n; ++x) acc += X[y * n + x] * C[v * n + x];
      tmp[y * n + v] = acc;
    }
  for (int u = 0; u < n; ++u)
    for (int v = 0; v < n; ++v) {
      float acc = 0.f;
      for (int y = 0; y < n; ++y) acc += C[u * n + y] * tmp[y * n + v];
      X[u * n + v] = acc;
    }
}

/* inverse (DCT-III, reference: src/nlkalman.c:307-360): X = C^T Y C */
static void dct2_inverse(const float *C, int n, float *Y, float *tmp) {
  for (int u = 0; u < n; ++u)
    for (int x = 0; x < n; ++x) {
      float acc = 0.f;
      for (int v = 0; v < n; ++v) acc += Y[u * n + v] * C[v * n + x];
      tmp[u * n + x] = acc;
    }
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      float acc = 0.f;
      for (int u = 0; u < n; ++u) acc += C[u * n + y] * tmp[u * n + x];
      Y[y * n + x] = acc;
    }
}

void nlko_dct2(float *planes, int n, int nplanes, int inverse) {
  float *C = malloc(sizeof(float) * n * n), *tmp = malloc(sizeof(float) * n * n);
  nlko_dct_basis(C, n);
  for (int i = 0; i < nplanes; ++i) {
    if (inverse) dct2_inverse(C, n, planes + (long)i * n * n, tmp);
    else dct2_forward(C, n, planes + (long)i * n * n, tmp);
  }
  free(C);
  free(tmp);
}

/* --------------------------------------------------------------- frame core */

typedef struct {
  float d;
  int idx; /* enumeration index in the window: qy-major, then qx */
  int x, y;
} cand_t;

/* ascending by distance, ties in window enumeration order: what the
 * reference's qsort (glibc: stable mergesort) yields for its comparator
 * (reference: src/nlkalman.c:500-505,706) */
static int cand_cmp(const void *a, const void *b) {
  const cand_t *p = a, *q = b;
  if (p->d < q->d) return -1;
  if (p->d > q->d) return 1;
  return (p->idx > q->idx) - (p->idx < q->idx);
}

typedef struct {
  int w, h, ch, psz, step, P2, E;
  float sigma2;
  nlko_params P;
  const float *cur;   /* frame whose patches are filtered (nisy1 / filt1) */
  const float *prev;  /* motion-compensated previous output, may be NULL */
  const float *match; /* matching + statistics image: bsic1 ? bsic1 : cur */
  int have_basic;
  float *out, *aggr;
  int *mask;
  float *W, *C;
  int smoother;
  int parallel; /* use atomics */
  nlko_trace *tr;
  int ngx, ngy;
} frame_ctx;

typedef struct {
  cand_t *cand;
  float *A, *B, *tmp;                   /* one patch: E coefficients each */
  float *M0, *M0V, *V0, *V01, *M1, *V1; /* statistics, E each */
  float *G, *G0;                        /* group slots [ntagg][E] */
  int *gx, *gy;
} scratch_t;

static scratch_t scratch_new(const frame_ctx *f) {
  scratch_t s;
  const int wmax = 2 * imax(f->P.search_sz_x, f->P.search_sz_t) + 1;
  const int ns = imax(f->P.npatches_tagg, 1);
  s.cand = malloc(sizeof(cand_t) * wmax * wmax);
  float *blk = malloc(sizeof(float) * f->E * (9 + 2 * (long)ns));
  s.A = blk;
  s.B = s.A + f->E;
  s.tmp = s.B + f->E;
  s.M0 = s.tmp + f->E;
  s.M0V = s.M0 + f->E;
  s.V0 = s.M0V + f->E;
  s.V01 = s.V0 + f->E;
  s.M1 = s.V01 + f->E;
  s.V1 = s.M1 + f->E;
  s.G = s.V1 + f->E;
  s.G0 = s.G + (long)ns * f->E;
  s.gx = malloc(sizeof(int) * ns * 2);
  s.gy = s.gx + ns;
  return s;
}
static void scratch_free(scratch_t *s) {
  free(s->cand);
  free(s->A);
  free(s->gx);
}

/* no NaN in channel 0 of the psz x psz patch of `im` at (qx,qy)
 * (reference: src/nlkalman.c:605-609, 725-730) */
static int patch_valid(const frame_ctx *f, const float *im, int qx, int qy) {
  if (!im) return 0;
  for (int hy = 0; hy < f->psz; ++hy)
    for (int hx = 0; hx < f->psz; ++hx)
      if (isnan(im[((long)(qx + hx) + (long)(qy + hy) * f->w) * f->ch])) return 0;
  return 1;
}

/* gather a patch into planar [c][hy][hx] order */
static void patch_load(const frame_ctx *f, const float *im, int qx, int qy,
                       float *dst) {
  for (int c = 0; c < f->ch; ++c)
    for (int hy = 0; hy < f->psz; ++hy)
      for (int hx = 0; hx < f->psz; ++hx)
        dst[c * f->P2 + hy * f->psz + hx] =
            im[((long)(qx + hx) + (long)(qy + hy) * f->w) * f->ch + c];
}

static void patch_dct(const frame_ctx *f, float *p, float *tmp, int inverse) {
  for (int c = 0; c < f->ch; ++c) {
    if (inverse) dct2_inverse(f->C, f->psz, p + c * f->P2, tmp);
    else dct2_forward(f->C, f->psz, p + c * f->P2, tmp);
  }
}

/* exhaustive block matching in the clamped window, then ranked selection
 * (reference: src/nlkalman.c:637-707). Returns the number of kept candidates;
 * s->cand[0..k) are sorted. Accumulation order hy -> hx -> c, one rounding per
 * multiply and per add (the file is built with -ffp-contract=off). */
static int block_match(const frame_ctx *f, scratch_t *s, int px, int py, int wsz,
                       int k) {
  const int x0 = imax(px - wsz, 0), x1 = imin(px + wsz, f->w - f->psz) + 1;
  const int y0 = imax(py - wsz, 0), y1 = imin(py + wsz, f->h - f->psz) + 1;
  const float norm = (float)f->psz * f->psz * f->ch;
  int n = 0;
  for (int qy = y0; qy < y1; ++qy)
    for (int qx = x0; qx < x1; ++qx, ++n) {
      float ww = 0;
      for (int hy = 0; hy < f->psz; ++hy) {
        const float *q = f->match + ((long)qx + (long)(qy + hy) * f->w) * f->ch;
        const float *t = f->match + ((long)px + (long)(py + hy) * f->w) * f->ch;
        for (int i = 0; i < f->psz * f->ch; ++i) {
          const float e = q[i] - t[i];
          ww += e * e;
        }
      }
      s->cand[n].x = qx;
      s->cand[n].y = qy;
      s->cand[n].idx = n;
      s->cand[n].d = fmax_ref(ww / norm, 0.f);
    }
  qsort(s->cand, n, sizeof(cand_t), cand_cmp);
  return imin(k, n);
}

static void add_at(const frame_ctx *f, float *p, float v) {
  if (f->parallel) {
#pragma omp atomic
    *p += v;
  } else
    *p += v;
}

/* weighted aggregation of the first nagg group slots
 * (reference: src/nlkalman.c:909-932, 1821-1845) */
static void aggregate(const frame_ctx *f, const scratch_t *s, int nagg, float vp,
                      int mark) {
  const float wgt = 1.f / fmax_ref(vp, 1e-6);
  for (int n = 0; n < nagg; ++n) {
    const int qx = s->gx[n], qy = s->gy[n];
    for (int hy = 0; hy < f->psz; ++hy)
      for (int hx = 0; hx < f->psz; ++hx) {
        const long pix = (long)(qx + hx) + (long)(qy + hy) * f->w;
        const float ww = wgt * f->W[hy * f->psz + hx];
        add_at(f, f->aggr + pix, ww);
        for (int c = 0; c < f->ch; ++c)
          add_at(f, f->out + pix * f->ch + c,
                 ww * s->G[(long)n * f->E + c * f->P2 + hy * f->psz + hx]);
      }
    if (f->parallel) {
#pragma omp atomic
      f->mask[qx + (long)qy * f->w] += mark;
    } else
      f->mask[qx + (long)qy * f->w] += mark;
  }
}

static void trace_target(const frame_ctx *f, const scratch_t *s, int gi, int k,
                         int np0, int np1, int nagg, float vp, int active) {
  nlko_trace *t = f->tr;
  if (!t) return;
  if (t->active) t->active[gi] = active;
  if (!active) return;
  if (t->nsel) t->nsel[gi] = k;
  if (t->np0) t->np0[gi] = np0;
  if (t->np1) t->np1[gi] = np1;
  if (t->nagg) t->nagg[gi] = nagg;
  if (t->vp) t->vp[gi] = vp;
  if (t->topk)
    for (int i = 0; i < t->kmax; ++i)
      t->topk[(long)gi * t->kmax + i] =
          i < k ? (s->cand[i].x | (s->cand[i].y << 16)) : -1;
  if (t->gcoords)
    for (int i = 0; i < t->gmax; ++i)
      t->gcoords[(long)gi * t->gmax + i] =
          i < nagg ? (s->gx[i] | (s->gy[i] << 16)) : -1;
}

/* one target patch of nlkalman_filter_frame (reference: src/nlkalman.c:597-932) */
static void filter_target(const frame_ctx *f, scratch_t *s, int px, int py, int gi) {
  const int E = f->E, ntagg = f->P.npatches_tagg;
  const float s2 = f->sigma2;
  const int prev_p = patch_valid(f, f->prev, px, py);
  int k = prev_p ? f->P.npatches_t : f->P.npatches_x;
  int np0 = 0, np1 = 0;
  for (int e = 0; e < E; ++e)
    s->M0[e] = s->M0V[e] = s->V0[e] = s->V01[e] = s->M1[e] = s->V1[e] = 0.f;

  if (k > 1) {
    const int wsz = prev_p ? f->P.search_sz_t : f->P.search_sz_x;
    k = block_match(f, s, px, py, wsz, k);
    /* Welford statistics over the kept candidates (reference: :713-811) */
    for (int i = 0; i < k; ++i) {
      const int qx = s->cand[i].x, qy = s->cand[i].y;
      const int prev = prev_p && patch_valid(f, f->prev, qx, qy);
      patch_load(f, f->match, qx, qy, s->A);
      patch_dct(f, s->A, s->tmp, 0);
      if (prev) {
        patch_load(f, f->prev, qx, qy, s->B);
        patch_dct(f, s->B, s->tmp, 0);
      }
      np1++;
      np0 += prev;
      const float inp1 = 1. / (float)np1;
      const float inp0 = prev ? 1. / (float)np0 : 0;
      const int slot = prev ? (np0 <= ntagg ? np0 - 1 : -1)
                            : (np1 <= ntagg ? np1 - 1 : -1);
      for (int e = 0; e < E; ++e) {
        const float a = s->A[e];
        const float d1 = a - s->M1[e];
        s->M1[e] += d1 * inp1;
        s->V1[e] += d1 * (a - s->M1[e]);
        if (prev) {
          const float b = s->B[e];
          const float d0 = b - s->M0V[e];
          s->M0V[e] += d0 * inp0;
          s->V0[e] += d0 * (b - s->M0V[e]);
          const float t = b - a;
          s->V01[e] += t * t;
          if (slot >= 0) s->M0[e] += (b - s->M0[e]) * inp0;
        }
      }
      if (slot >= 0) {
        s->gx[slot] = qx;
        s->gy[slot] = qy;
        if (f->have_basic) patch_load(f, f->cur, qx, qy, s->G + (long)slot * E);
        else memcpy(s->G + (long)slot * E, s->A, sizeof(float) * E);
      }
    }
    const float inp1 = 1. / (float)np1, inp0 = np0 ? 1. / (float)np0 : 0;
    for (int e = 0; e < E; ++e) {
      s->V1[e] *= inp1;
      if (np0) {
        s->V0[e] *= inp0;
        s->V01[e] *= inp0;
      }
    }
  } else {
    /* "local" mode (reference: :815-849): point estimates; np0 = np1 = 0 so
     * nothing is aggregated below — kept for completeness */
    k = 0;
  }

  /* second iteration: the group slots hold noisy pixels (reference: :853) */
  const int nagg = imin(np0 ? np0 : np1, ntagg);
  if (f->have_basic)
    for (int n = 0; n < nagg; ++n) patch_dct(f, s->G + (long)n * E, s->tmp, 0);

  /* Kalman / Wiener shrinkage (reference: :855-904) */
  float vp = 0;
  for (int n = 0; n < nagg; ++n) {
    float *g = s->G + (long)n * E;
    if (np0 > 0) {
      for (int e = 0; e < E; ++e) {
        const float v = s->V0[e] + fmax_ref(0.f, s->V01[e] - (f->have_basic ? 0 : s2));
        const float a = v / (v + f->P.beta_t * s2);
        vp += (1 - a * a) * v + a * a * s2;
        g[e] = a * g[e] + (1 - a) * s->M0[e];
      }
    } else {
      for (int e = 0; e < E; ++e) {
        const float v = fmax_ref(0.f, s->V1[e] - (f->have_basic ? 0 : s2));
        const float a = v / (v + f->P.beta_x * s2);
        vp += a * v;
        g[e] = a * g[e] + (1 - a) * s->M1[e];
      }
    }
    patch_dct(f, g, s->tmp, 1); /* reference: :906 */
  }
  trace_target(f, s, gi, k, np0, np1, nagg, vp, 1);
  /* groups of a temporal frame without any valid previous patch do not mark
   * the processed-mask (reference: :931) */
  aggregate(f, s, nagg, vp, (f->prev && !np0) ? 0 : 1);
}

/* one target patch of nlkalman_smooth_frame (reference: src/nlkalman.c:1490-1845) */
static void smooth_target(const frame_ctx *f, scratch_t *s, int px, int py, int gi) {
  const int E = f->E, ntagg = f->P.npatches_tagg;
  const int prev_p = patch_valid(f, f->prev, px, py);
  int k = prev_p ? f->P.npatches_t : f->P.npatches_x;
  int np0 = 0, np1 = 0;
  for (int e = 0; e < E; ++e)
    s->M0[e] = s->V0[e] = s->V01[e] = s->M1[e] = s->V1[e] = 0.f;

  if (k > 1) {
    k = block_match(f, s, px, py, f->P.search_sz_t, k); /* reference: :1527 */
    for (int i = 0; i < k; ++i) { /* reference: :1603-1680 */
      const int qx = s->cand[i].x, qy = s->cand[i].y;
      const int prev = prev_p && patch_valid(f, f->prev, qx, qy);
      patch_load(f, f->match, qx, qy, s->A);
      patch_dct(f, s->A, s->tmp, 0);
      if (prev) {
        patch_load(f, f->prev, qx, qy, s->B);
        patch_dct(f, s->B, s->tmp, 0);
      }
      np1++;
      np0 += prev;
      const float inp1 = 1. / (float)np1;
      const float inp0 = prev ? 1. / (float)np0 : 0;
      const int slot = (prev && np0 <= ntagg) ? np0 - 1 : -1;
      for (int e = 0; e < E; ++e) {
        const float a = s->A[e];
        const float d1 = a - s->M1[e];
        s->M1[e] += d1 * inp1;
        s->V1[e] += d1 * (a - s->M1[e]);
        if (prev) {
          const float b = s->B[e];
          const float d0 = b - s->M0[e];
          s->M0[e] += d0 * inp0;
          s->V0[e] += d0 * (b - s->M0[e]);
          const float t = b - a;
          s->V01[e] += t * t;
        }
      }
      if (slot >= 0) {
        s->gx[slot] = qx;
        s->gy[slot] = qy;
        memcpy(s->G0 + (long)slot * E, s->B, sizeof(float) * E);
        if (f->have_basic) patch_load(f, f->cur, qx, qy, s->G + (long)slot * E);
        else memcpy(s->G + (long)slot * E, s->A, sizeof(float) * E);
      }
    }
    const float inp1 = 1. / (float)np1, inp0 = np0 ? 1. / (float)np0 : 0;
    for (int e = 0; e < E; ++e) {
      s->V1[e] *= inp1;
      if (np0) {
        s->V0[e] *= inp0;
        s->V01[e] *= inp0;
      }
    }
  } else {
    /* the reference's single-patch mode for the smoother (:1699-1730) reads an
     * unset group coordinate and mis-indexes F1S0 (:1727); it is unreachable
     * with npatches_t >= 2. Here such a target is passed through like a
     * target without a valid previous patch. */
    k = 0;
  }

  int nagg = imin(np0, ntagg);
  if (f->have_basic)
    for (int n = 0; n < nagg; ++n) patch_dct(f, s->G + (long)n * E, s->tmp, 0);

  float vp = 0;
  const float b = f->P.beta_t;
  for (int n = 0; n < nagg; ++n) { /* reference: :1739-1777, :1793 */
    float *g1 = s->G + (long)n * E;
    const float *g0 = s->G0 + (long)n * E;
    for (int e = 0; e < E; ++e) {
      const float a = s->V1[e] / (s->V1[e] + b * s->V01[e]);
      vp += (1 - a * a) * s->V1[e] + a * a * fmax_ref(s->V0[e] - b * s->V01[e], 0.f);
      g1[e] = (1 - a) * g1[e] + a * g0[e];
    }
    patch_dct(f, g1, s->tmp, 1);
  }
  if (np0 == 0) { /* reference: :1795-1804 — pass the target patch through */
    nagg = 1;
    s->gx[0] = px;
    s->gy[0] = py;
    patch_load(f, f->cur, px, py, s->G);
  }
  trace_target(f, s, gi, k, np0, np1, nagg, vp, 1);
  aggregate(f, s, nagg, vp, np0 ? 1 : 0); /* reference: :1844 */
}

/* When non-NULL, the processed-mask test is replaced by these externally decided
 * flags (one per strip target, 1 = process): the "group" phase of the three-phase
 * strip form below. */
static const unsigned char *g_active_in = NULL;

/* oy/ngy: first image row and number of patch-grid rows to process (a whole
 * frame is oy = 0, ngy = (h - psz)/step + 1). acc != NULL selects the row-strip
 * form: nothing is normalised, the weighted sums and weights are ADDED to the
 * planar accumulator acc[(ch+1)][h][w] (mirror of nlk_dev_frame_accumulate). */
static void run_frame(float *out, const float *cur, const float *prev,
                      const float *basic, int w, int h, int ch, float sigma,
                      const nlko_params *P, int nthreads, nlko_trace *tr,
                      int smoother, int oy, int ngy, float *acc) {
  frame_ctx f;
  memset(&f, 0, sizeof f);
  f.w = w; f.h = h; f.ch = ch;
  f.psz = P->patch_sz;
  f.step = f.psz / 2;
  f.P2 = f.psz * f.psz;
  f.E = f.P2 * ch;
  f.sigma2 = sigma * sigma;
  f.P = *P;
  f.cur = cur; f.prev = prev;
  f.match = basic ? basic : cur;
  f.have_basic = basic != NULL;
  f.out = out;
  f.smoother = smoother;
  f.tr = tr;
  f.aggr = calloc((size_t)w * h, sizeof(float));
  f.mask = calloc((size_t)w * h, sizeof(int));
  f.W = malloc(sizeof(float) * f.P2);
  f.C = malloc(sizeof(float) * f.P2);
  nlko_window(f.W, f.psz);
  nlko_dct_basis(f.C, f.psz);
  memset(out, 0, sizeof(float) * (size_t)w * h * ch);
  /* the reference's loops `for (px = 0; px < w - psz + 1; px += step)` (:586, :595): no target at all in an
   * image smaller than a patch, and the output is then the input (:939-942) */
  f.ngx = w >= f.psz ? (w - f.psz) / f.step + 1 : 0;
  f.ngy = ngy;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 100) nthreads = 100; /* the reference aborts above 100 (:165) */
#else
  nthreads = 1;
#endif
  f.parallel = nthreads > 1;

  /* raster loop over the patch grid with the processed-mask skip
   * (reference: src/nlkalman.c:586-600, 1477-1493); rows are split statically
   * over threads exactly as the reference's `omp parallel for` does */
#pragma omp parallel num_threads(nthreads) if (nthreads > 1)
  {
    scratch_t s = scratch_new(&f);
#pragma omp for schedule(static)
    for (int gy = 0; gy < f.ngy; ++gy)
      for (int gx = 0; gx < f.ngx; ++gx) {
        const int px = gx * f.step, py = oy + gy * f.step;
        int m;
#pragma omp atomic read
        m = f.mask[px + (long)py * w];
        if (g_active_in) m = !g_active_in[gx + gy * f.ngx];
        if (m) {
          if (tr && tr->active) tr->active[gx + gy * f.ngx] = 0;
          continue;
        }
        if (smoother) smooth_target(&f, &s, px, py, gx + gy * f.ngx);
        else filter_target(&f, &s, px, py, gx + gy * f.ngx);
      }
    scratch_free(&s);
  }

  if (acc) {
    const long npix = (long)w * h;
    for (long i = 0; i < npix; ++i) {
      for (int c = 0; c < ch; ++c) acc[c * npix + i] += out[i * ch + c];
      acc[ch * npix + i] += f.aggr[i];
    }
  } else
  /* normalisation with the absolute 1e-6 threshold (reference: :939-942, :1853-1856) */
  for (long i = 0; i < (long)w * h; ++i)
    for (int c = 0; c < ch; ++c) {
      if (f.aggr[i] > 1e-6) out[i * ch + c] /= f.aggr[i];
      else out[i * ch + c] = cur[i * ch + c];
    }
  if (tr && tr->aggr) memcpy(tr->aggr, f.aggr, sizeof(float) * (size_t)w * h);
  free(f.aggr); free(f.mask); free(f.W); free(f.C);
}

/* reference: src/nlkalman.c:518-951 */
void nlko_filter_frame(float *deno1, const float *nisy1, const float *deno0,
                       const float *bsic1, int w, int h, int ch, float sigma,
                       const nlko_params *P, int nthreads, nlko_trace *tr) {
  run_frame(deno1, nisy1, deno0, bsic1, w, h, ch, sigma, P, nthreads, tr, 0, 0,
            h >= P->patch_sz ? (h - P->patch_sz) / (P->patch_sz / 2) + 1 : 0, NULL);
}

/* reference: src/nlkalman.c:1409-1865 */
void nlko_smooth_frame(float *smoo1, const float *filt1, const float *smoo0,
                       const float *bsic1, int w, int h, int ch, float sigma,
                       const nlko_params *P, int nthreads, nlko_trace *tr) {
  run_frame(smoo1, filt1, smoo0, bsic1, w, h, ch, sigma, P, nthreads, tr, 1, 0,
            h >= P->patch_sz ? (h - P->patch_sz) / (P->patch_sz / 2) + 1 : 0, NULL);
}

/* row-strip form (mirror of nlk_dev_frame_accumulate / _normalize in include/nlk_hip.h) */
void nlko_frame_accumulate(float *acc, const float *cur, const float *prev,
                           const float *basic, int w, int h, int ch, float sigma,
                           const nlko_params *P, int oy, int ngy, int smoother) {
  float *tmp = malloc(sizeof(float) * (size_t)w * h * ch);
  run_frame(tmp, cur, prev, basic, w, h, ch, sigma, P, 1, NULL, smoother, oy, ngy, acc);
  free(tmp);
}

void nlko_frame_normalize(float *out, const float *acc, const float *cur, int w, int h,
                          int ch, int y0, int y1) {
  const long npix = (long)w * h;
  for (long i = (long)y0 * w; i < (long)y1 * w; ++i)
    for (int c = 0; c < ch; ++c)
      out[i * ch + c] = acc[ch * npix + i] > 1e-6 ? acc[c * npix + i] / acc[ch * npix + i]
                                                  : cur[i * ch + c];
}

/* ---- three-phase strip form (mirror of nlk_dev_strip_match / nlk_dev_mask_commit /
 * nlk_dev_strip_group in include/nlk_hip.h) */

/* Phase 1: for EVERY target of the strip, the group it would aggregate
 * (reference: src/nlkalman.c:605-609, 630-707, 725-732, 779-793, 857) encoded as a
 * mark word: bit (dj+R)*(2R+1) + (di+R) is set when the group holds the grid target
 * (di, dj) away and the group marks the processed-mask (reference: :931, :1844). */
void nlko_strip_match(uint64_t *marks, const float *cur, const float *prev,
                      const float *basic, int w, int h, int ch, float sigma,
                      const nlko_params *P, int oy, int ngy, int smoother, int *reach) {
  frame_ctx f;
  memset(&f, 0, sizeof f);
  (void)sigma;
  f.w = w; f.h = h; f.ch = ch;
  f.psz = P->patch_sz; f.step = f.psz / 2; f.P2 = f.psz * f.psz; f.E = f.P2 * ch;
  f.P = *P;
  f.cur = cur; f.prev = prev; f.match = basic ? basic : cur; f.have_basic = basic != NULL;
  f.ngx = (w - f.psz) / f.step + 1;
  f.ngy = ngy;
  const int wmark = (smoother || prev) ? P->search_sz_t : P->search_sz_x;
  const int R = wmark / f.step, side = 2 * R + 1;
  if (reach) *reach = R;
  scratch_t s = scratch_new(&f);
  const int ntagg = P->npatches_tagg;
  for (int gy = 0; gy < ngy; ++gy)
    for (int gx = 0; gx < f.ngx; ++gx) {
      const int px = gx * f.step, py = oy + gy * f.step;
      uint64_t m = 0;
      const int prev_p = patch_valid(&f, prev, px, py);
      int k = prev_p ? P->npatches_t : P->npatches_x;
      if (k > 1) {
        const int wsz = (smoother || prev_p) ? P->search_sz_t : P->search_sz_x;
        k = block_match(&f, &s, px, py, wsz, k);
        int np0 = 0, np1 = 0;
        for (int i = 0; i < k; ++i) {
          const int pv = prev_p && patch_valid(&f, prev, s.cand[i].x, s.cand[i].y);
          np1++;
          np0 += pv;
          const int slot = pv ? (np0 <= ntagg ? np0 - 1 : -1)
                              : ((!smoother && np1 <= ntagg) ? np1 - 1 : -1);
          if (slot >= 0) { s.gx[slot] = s.cand[i].x; s.gy[slot] = s.cand[i].y; }
        }
        const int nagg = smoother ? imin(np0, ntagg) : imin(np0 ? np0 : np1, ntagg);
        const int mark = smoother ? (np0 > 0) : !(prev && np0 == 0);
        for (int n = 0; mark && n < nagg; ++n) {
          const int dx = s.gx[n] - px, dy = s.gy[n] - py;
          if (dx % f.step == 0 && dy % f.step == 0)
            m |= 1ull << ((dy / f.step + R) * side + dx / f.step + R);
        }
      }
      marks[gx + (long)gy * f.ngx] = m;
    }
  scratch_free(&s);
}

/* Phase 2: the raster-order replay itself (reference: src/nlkalman.c:597-600 skip,
 * :930-931 mark) on mark words: a target is processed iff no earlier processed
 * target's group contained it. */
void nlko_mask_commit(const uint64_t *marks, int ngx, int ngy, int R, unsigned char *active) {
  const int side = 2 * R + 1;
  unsigned char *hit = calloc((size_t)ngx * ngy, 1);
  for (int j = 0; j < ngy; ++j)
    for (int i = 0; i < ngx; ++i) {
      const long t = i + (long)j * ngx;
      active[t] = !hit[t];
      if (hit[t]) continue;
      for (int b = 0; b < side * side; ++b)
        if ((marks[t] >> b) & 1) {
          const int jj = j + b / side - R, ii = i + b % side - R;
          if (jj >= 0 && jj < ngy && ii >= 0 && ii < ngx) hit[ii + (long)jj * ngx] = 1;
        }
    }
  free(hit);
}

/* Phase 3: filter + aggregate the strip's targets flagged in `active` */
void nlko_strip_group(float *acc, const unsigned char *active, const float *cur,
                      const float *prev, const float *basic, int w, int h, int ch,
                      float sigma, const nlko_params *P, int oy, int ngy, int smoother) {
  float *tmp = malloc(sizeof(float) * (size_t)w * h * ch);
  g_active_in = active;
  run_frame(tmp, cur, prev, basic, w, h, ch, sigma, P, 1, NULL, smoother, oy, ngy, acc);
  g_active_in = NULL;
  free(tmp);
}

int nlko_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
