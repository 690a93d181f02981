/* tvl1_oracle.c — CPU restatement of the reference's dual TV-L1 optical flow
 *  and of the occlusion mask the pipelines derive from it.
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg, never by the product path.
 *
 * PARITY PINNED: unlike the filter (which needs FFTW3), lib/tvl1flow compiles from
 * its own five C files, so oracle/Makefile builds it where it lies into
 * oracle/_ref/libtvl1flow_ref.so and tests/test_tvl1.py requires this restatement
 * to reproduce it BIT FOR BIT (single thread; both built with -ffp-contract=off).
 * Every expression below therefore keeps the reference's operand types and
 * association order (several of them mix float and double), cited per function.
 *
 * Layout: row-major float images, x fastest. Flow: u1 = x component, u2 = y.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "nlk_oracle.h"

/* ---- sampling (reference: lib/tvl1flow/bicubic_interpolation.c:26-41, 100-131, 140-236) */

/* Neumann clamp; *out is raised when the index had to be moved */
static int clampi(int x, int n, int* out) {
  if (x < 0) { *out = 1; return 0; }
  if (x >= n) { *out = 1; return n - 1; }
  return x;
}

/* Catmull-Rom cell in double, Horner form of bicubic_interpolation.c:104-109 */
static double cubic4(const double v[4], double t) {
  return v[1] + 0.5 * t * (v[2] - v[0] +
         t * (2.0 * v[0] - 5.0 * v[1] + 4.0 * v[2] - v[3] +
         t * (3.0 * (v[1] - v[2]) + v[3] - v[0])));
}

float tvl1o_bicubic_at(const float* im, float uu, float vv, int nx, int ny, int border_out) {
  const int sx = uu < 0 ? -1 : 1, sy = vv < 0 ? -1 : 1;
  int out = 0;
  /* taps x-1, x, x+1, x+2 in the direction of the sign (truncation toward zero);
   * the row before y is offset by sx, not sy: bicubic_interpolation.c:157 does that */
  int cx[4], cy[4];
  cx[1] = clampi((int)uu, nx, &out);
  cy[1] = clampi((int)vv, ny, &out);
  cx[0] = clampi((int)uu - sx, nx, &out);
  cy[0] = clampi((int)vv - sx, ny, &out);
  cx[2] = clampi((int)uu + sx, nx, &out);
  cy[2] = clampi((int)vv + sy, ny, &out);
  cx[3] = clampi((int)uu + 2 * sx, nx, &out);
  cy[3] = clampi((int)vv + 2 * sy, ny, &out);
  if (out && border_out) return 0.0f;
  double col[4];
  for (int a = 0; a < 4; ++a) {  /* along y first, then along x (:122-131) */
    double tap[4];
    for (int b = 0; b < 4; ++b) tap[b] = im[cx[a] + nx * cy[b]];
    col[a] = cubic4(tap, vv - cy[1]);
  }
  return (float)cubic4(col, uu - cx[1]);
}

static void warp_flow(const float* im, const float* u, const float* v, float* out, int nx, int ny) {
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) {
      const int p = i * nx + j;  /* bicubic_interpolation.c:252-263, border_out = true */
      out[p] = tvl1o_bicubic_at(im, (float)(j + u[p]), (float)(i + v[p]), nx, ny, 1);
    }
}

/* ---- differential operators (reference: lib/tvl1flow/mask.c:43-214) */

void tvl1o_forward_gradient(const float* f, float* fx, float* fy, int nx, int ny) {
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) {
      const int p = i * nx + j;  /* mask.c:104-140: zero across the last column / row */
      fx[p] = j < nx - 1 ? f[p + 1] - f[p] : 0.0f;
      fy[p] = i < ny - 1 ? f[p + nx] - f[p] : 0.0f;
    }
}

void tvl1o_centered_gradient(const float* f, float* dx, float* dy, int nx, int ny) {
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) {
      const int k = i * nx + j;  /* mask.c:158-213: one-sided at the borders, still halved */
      const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
      const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
      dx[k] = (float)(0.5 * (f[i * nx + jr] - f[i * nx + jl]));
      dy[k] = (float)(0.5 * (f[id * nx + j] - f[iu * nx + j]));
    }
}

/* backward-difference divergence; the association order differs between the body,
 * the first/last column and the corners in the reference (mask.c:52-96) and is kept */
void tvl1o_divergence(const float* v1, const float* v2, float* div, int nx, int ny) {
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) {
      const int p = i * nx + j;
      const int top = i == 0, bot = i == ny - 1, lef = j == 0, rig = j == nx - 1;
      float d;
      if (!lef && !rig) {
        const float ax = v1[p] - v1[p - 1];
        if (!top && !bot) d = ax + (v2[p] - v2[p - nx]);    /* :59-65 */
        else if (top) d = ax + v2[p];                        /* :74 */
        else d = ax - v2[p - nx];                            /* :75 */
      } else if (!top && !bot) {
        if (lef) d = v1[p] + v2[p] - v2[p - nx];             /* :84 */
        else d = -v1[p - 1] + v2[p] - v2[p - nx];            /* :85 */
      } else if (top) {
        d = lef ? v1[p] + v2[p] : -v1[p - 1] + v2[p];        /* :89-90 */
      } else {
        d = lef ? v1[p] - v2[p - nx] : -v1[p - 1] - v2[p - nx]; /* :91-92 */
      }
      div[p] = d;
    }
}

/* ---- Gaussian (reference: lib/tvl1flow/mask.c:221-330, reflecting boundary, window 5 sigma) */

void tvl1o_gaussian(float* im, int nx, int ny, double sigma) {
  const int rad = (int)(5 * sigma) + 1;  /* taps 0 .. rad-1 on each side */
  if (rad > nx) abort();                 /* mask.c:238-241 */
  double* B = (double*)malloc(sizeof(double) * rad);
  const double den = 2 * sigma * sigma;
  for (int i = 0; i < rad; ++i) B[i] = 1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den);
  double norm = 0;
  for (int i = 0; i < rad; ++i) norm += B[i];
  norm *= 2;
  norm -= B[0];
  for (int i = 0; i < rad; ++i) B[i] /= norm;

  const int nmax = nx > ny ? nx : ny;
  double* line = (double*)malloc(sizeof(double) * (nmax + 2 * rad));
  for (int pass = 0; pass < 2; ++pass) {  /* rows, then columns of the row-filtered image */
    const int n = pass ? ny : nx, m = pass ? nx : ny, stride = pass ? nx : 1;
    for (int k = 0; k < m; ++k) {
      float* base = im + (pass ? k : k * nx);
      for (int i = 0; i < n; ++i) line[rad + i] = base[i * stride];
      /* the left pad mirrors about sample 0 without repeating it, the right pad repeats
       * the last sample (mask.c:271-276, 308-313) */
      for (int i = 0; i < rad; ++i) {
        line[i] = base[(rad - i) * stride];
        line[rad + n + i] = base[(n - i - 1) * stride];
      }
      for (int i = rad; i < rad + n; ++i) {
        double sum = B[0] * line[i];
        for (int j = 1; j < rad; ++j) sum += B[j] * (line[i - j] + line[i + j]);
        base[(i - rad) * stride] = (float)sum;
      }
    }
  }
  free(line);
  free(B);
}

/* ---- pyramid (reference: lib/tvl1flow/zoom.c:23-108) */

void tvl1o_zoom_size(int nx, int ny, int* nxx, int* nyy, float factor) {
  *nxx = (int)((float)nx * factor + 0.5);
  *nyy = (int)((float)ny * factor + 0.5);
}

void tvl1o_zoom_out(const float* im, float* out, int nx, int ny, float factor) {
  float* s = (float*)malloc(sizeof(float) * nx * ny);
  memcpy(s, im, sizeof(float) * nx * ny);
  int nxx, nyy;
  tvl1o_zoom_size(nx, ny, &nxx, &nyy, factor);
  const float sigma = 0.6 * sqrt(1.0 / (factor * factor) - 1.0);  /* zoom.c:60 */
  tvl1o_gaussian(s, nx, ny, sigma);
  for (int i1 = 0; i1 < nyy; ++i1)
    for (int j1 = 0; j1 < nxx; ++j1)
      out[i1 * nxx + j1] = tvl1o_bicubic_at(s, (float)j1 / factor, (float)i1 / factor, nx, ny, 0);
  free(s);
}

void tvl1o_zoom_in(const float* im, float* out, int nx, int ny, int nxx, int nyy) {
  const float fx = (float)nxx / nx, fy = (float)nyy / ny;  /* zoom.c:94-95 */
  for (int i1 = 0; i1 < nyy; ++i1)
    for (int j1 = 0; j1 < nxx; ++j1)
      out[i1 * nxx + j1] = tvl1o_bicubic_at(im, (float)j1 / fx, (float)i1 / fy, nx, ny, 0);
}

/* ---- one scale (reference: lib/tvl1flow/tvl1flow_lib.c:93-275) */

int tvl1o_flow_scale(const float* I0, const float* I1, float* u1, float* u2, int nx, int ny, float tau,
                     float lambda, float theta, int warps, float epsilon, int* iters_out) {
  const int size = nx * ny;
  const float l_t = lambda * theta;
  float* buf = (float*)malloc(sizeof(float) * (size_t)size * 17);
  float *I1x = buf, *I1y = I1x + size, *I1w = I1y + size, *I1wx = I1w + size, *I1wy = I1wx + size;
  float *rho_c = I1wy + size, *grad = rho_c + size, *v1 = grad + size, *v2 = v1 + size;
  float *p11 = v2 + size, *p12 = p11 + size, *p21 = p12 + size, *p22 = p21 + size;
  float *dv1 = p22 + size, *dv2 = dv1 + size, *gx = dv2 + size, *gy = gx + size;
  int total = 0;
  tvl1o_centered_gradient(I1, I1x, I1y, nx, ny);
  memset(p11, 0, sizeof(float) * (size_t)size * 4);
  for (int wi = 0; wi < warps; ++wi) {
    warp_flow(I1, u1, u2, I1w, nx, ny);
    warp_flow(I1x, u1, u2, I1wx, nx, ny);
    warp_flow(I1y, u1, u2, I1wy, nx, ny);
    for (int i = 0; i < size; ++i) {  /* :151-162 */
      const float Ix2 = I1wx[i] * I1wx[i], Iy2 = I1wy[i] * I1wy[i];
      grad[i] = Ix2 + Iy2;
      rho_c[i] = I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i];
    }
    int n = 0;
    float error = INFINITY;
    while (error > epsilon * epsilon && n < 300) {  /* MAX_ITERATIONS, :24, :166 */
      ++n;
      for (int i = 0; i < size; ++i) {  /* thresholding step TH, :172-208 */
        const float rho = rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
        float d1, d2;
        if (rho < -l_t * grad[i]) {
          d1 = l_t * I1wx[i];
          d2 = l_t * I1wy[i];
        } else if (rho > l_t * grad[i]) {
          d1 = -l_t * I1wx[i];
          d2 = -l_t * I1wy[i];
        } else if (grad[i] < 1E-10) {
          d1 = d2 = 0;
        } else {
          const float fi = -rho / grad[i];
          d1 = fi * I1wx[i];
          d2 = fi * I1wy[i];
        }
        v1[i] = u1[i] + d1;
        v2[i] = u2[i] + d2;
      }
      tvl1o_divergence(p11, p12, dv1, nx, ny);
      tvl1o_divergence(p21, p22, dv2, nx, ny);
      error = 0.0f;
      for (int i = 0; i < size; ++i) {  /* :217-230 */
        const float a = u1[i], b = u2[i];
        u1[i] = v1[i] + theta * dv1[i];
        u2[i] = v2[i] + theta * dv2[i];
        error += (u1[i] - a) * (u1[i] - a) + (u2[i] - b) * (u2[i] - b);
      }
      error /= size;
      /* dual update, :233-250: hypot and the 1 + taut*g sum are evaluated in double */
      const float taut = tau / theta;
      for (int c = 0; c < 2; ++c) {
        float* pa = c ? p21 : p11;
        float* pb = c ? p22 : p12;
        tvl1o_forward_gradient(c ? u2 : u1, gx, gy, nx, ny);
        for (int i = 0; i < size; ++i) {
          const float g = hypot(gx[i], gy[i]);
          const float ng = 1.0 + taut * g;
          pa[i] = (pa[i] + taut * gx[i]) / ng;
          pb[i] = (pb[i] + taut * gy[i]) / ng;
        }
      }
    }
    total += n;
    if (iters_out) iters_out[wi] = n;
  }
  free(buf);
  return total;
}

/* ---- multiscale driver (reference: lib/tvl1flow/tvl1flow_lib.c:283-474) */

void tvl1o_normalize(const float* I0, const float* I1, float* o0, float* o1, int n) {
  float lo = I0[0], hi = I0[0];
  for (int i = 0; i < n; ++i) { lo = I0[i] < lo ? I0[i] : lo; hi = I0[i] > hi ? I0[i] : hi; }
  for (int i = 0; i < n; ++i) { lo = I1[i] < lo ? I1[i] : lo; hi = I1[i] > hi ? I1[i] : hi; }
  const float den = hi - lo;
  for (int i = 0; i < n; ++i) {  /* :320-333 (255.0 makes the scaling double) */
    o0[i] = den > 0 ? (float)(255.0 * (I0[i] - lo) / den) : I0[i];
    o1[i] = den > 0 ? (float)(255.0 * (I1[i] - lo) / den) : I1[i];
  }
}

/* the number of scales the command line derives from the image size (main.c:152-157) */
int tvl1o_auto_scales(int nx, int ny, int nscales, float zfactor) {
  const float N = 1 + log(hypot(nx, ny) / 16.0) / log(1 / zfactor);
  return N < nscales ? (int)N : nscales;
}

void tvl1o_flow(const float* I0, const float* I1, float* u1, float* u2, int nx, int ny, float tau,
                float lambda, float theta, int nscales, int fscale, float zfactor, int warps,
                float epsilon) {
  float **P0 = malloc(sizeof(float*) * nscales), **P1 = malloc(sizeof(float*) * nscales);
  float **U1 = malloc(sizeof(float*) * nscales), **U2 = malloc(sizeof(float*) * nscales);
  int *W = malloc(sizeof(int) * nscales), *H = malloc(sizeof(int) * nscales);
  W[0] = nx; H[0] = ny;
  P0[0] = malloc(sizeof(float) * nx * ny);
  P1[0] = malloc(sizeof(float) * nx * ny);
  U1[0] = u1; U2[0] = u2;
  tvl1o_normalize(I0, I1, P0[0], P1[0], nx * ny);
  tvl1o_gaussian(P0[0], nx, ny, 0.8);  /* PRESMOOTHING_SIGMA, :23, :380-381 */
  tvl1o_gaussian(P1[0], nx, ny, 0.8);
  for (int s = 1; s < nscales; ++s) {
    tvl1o_zoom_size(W[s - 1], H[s - 1], &W[s], &H[s], zfactor);
    const size_t n = (size_t)W[s] * H[s];
    P0[s] = malloc(sizeof(float) * n); P1[s] = malloc(sizeof(float) * n);
    U1[s] = malloc(sizeof(float) * n); U2[s] = malloc(sizeof(float) * n);
    tvl1o_zoom_out(P0[s - 1], P0[s], W[s - 1], H[s - 1], zfactor);
    tvl1o_zoom_out(P1[s - 1], P1[s], W[s - 1], H[s - 1], zfactor);
  }
  {
    const size_t n = (size_t)W[nscales - 1] * H[nscales - 1];
    memset(U1[nscales - 1], 0, sizeof(float) * n);
    memset(U2[nscales - 1], 0, sizeof(float) * n);
  }
  /* coarse to fine; scales finer than fscale only receive the upsampled flow (:418-461) */
  for (int s = nscales - 1; s >= 0; --s) {
    if (s >= fscale)
      tvl1o_flow_scale(P0[s], P1[s], U1[s], U2[s], W[s], H[s], tau, lambda, theta, warps, epsilon, NULL);
    if (s == 0) break;
    tvl1o_zoom_in(U1[s], U1[s - 1], W[s], H[s], W[s - 1], H[s - 1]);
    tvl1o_zoom_in(U2[s], U2[s - 1], W[s], H[s], W[s - 1], H[s - 1]);
    const size_t n = (size_t)W[s - 1] * H[s - 1];
    for (size_t i = 0; i < n; ++i) {
      U1[s - 1][i] *= (float)1.0 / zfactor;
      U2[s - 1][i] *= (float)1.0 / zfactor;
    }
  }
  for (int s = 0; s < nscales; ++s) {
    free(P0[s]); free(P1[s]);
    if (s) { free(U1[s]); free(U2[s]); }
  }
  free(P0); free(P1); free(U1); free(U2); free(W); free(H);
}

/* ---- occlusion mask of the pipelines (reference: scripts/nlkalman-seq.sh:70-73: plambda
 * "x(0,0)[0] x(-1,0)[0] - x(0,0)[1] x(0,-1)[1] - + fabs TH > 255 *", float stack, nearest
 * boundary extension) on an interleaved 2-channel flow */
void tvl1o_occlusion_mask(const float* flow, float* mask, int nx, int ny, float th) {
  for (int i = 0; i < ny; ++i)
    for (int j = 0; j < nx; ++j) {
      const int p = i * nx + j, pl = i * nx + (j > 0 ? j - 1 : 0), pu = (i > 0 ? i - 1 : 0) * nx + j;
      const float a = flow[2 * p] - flow[2 * pl], b = flow[2 * p + 1] - flow[2 * pu + 1];
      mask[p] = (fabsf(a + b) > th ? 1.0f : 0.0f) * 255.0f;
    }
}
