// k_tvl1.h — dual TV-L1 optical flow (Zach-Pock-Bischof) and the pipelines' occlusion
// mask, SURVEY.md §8(f-3) (reference: lib/tvl1flow/tvl1flow_lib.c, mask.c, zoom.c,
// bicubic_interpolation.c; scripts/nlkalman-seq.sh:70-73).
//
// All kernels are streaming / small-stencil passes over planar float images (x fastest),
// one thread per pixel, bound by HBM / Infinity-Cache bandwidth; there is no matrix-shaped
// work. The reference mixes float and double freely (double Gaussian sums, double Catmull-
// Rom cells, double hypot); every expression keeps its operand types and association order
// and contraction is off, so the results equal the CPU oracle's bit for bit as long as the
// solver stops at the same iteration (the only reordered float sum is the convergence
// measure: see k_tv_dual).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NLK_TV_MAXIT 300  // reference: tvl1flow_lib.c:24

// device-side state of the fixed-point loop of one warp
struct NlkTvState {
  int stop_iter;   // iterations n > stop_iter are no-ops (set by the iteration that converged)
  int iters;       // last executed iteration
  float error;     // its mean squared update
};

// ---- sampling (reference: bicubic_interpolation.c:26-41, 100-131, 140-236)
__device__ __forceinline__ int nlk_tv_clamp(int x, int n, bool& out) {
  if (x < 0) { out = true; return 0; }
  if (x >= n) { out = true; return n - 1; }
  return x;
}

__device__ __forceinline__ double nlk_tv_cubic(const double (&v)[4], double t) {
#pragma clang fp contract(off)
  return v[1] + 0.5 * t * (v[2] - v[0] +
         t * (2.0 * v[0] - 5.0 * v[1] + 4.0 * v[2] - v[3] +
         t * (3.0 * (v[1] - v[2]) + v[3] - v[0])));
}

// taps and weights of one sample position, shared by the images warped with the same flow
struct NlkTvTaps {
  int cx[4], cy[4];
  double tx, ty;
  bool out;
};

__device__ __forceinline__ NlkTvTaps nlk_tv_taps(float uu, float vv, int nx, int ny) {
  NlkTvTaps t;
  const int sx = uu < 0 ? -1 : 1, sy = vv < 0 ? -1 : 1;
  t.out = false;
  // (the row before y is offset by sx, not sy: bicubic_interpolation.c:157)
  t.cx[1] = nlk_tv_clamp((int)uu, nx, t.out);
  t.cy[1] = nlk_tv_clamp((int)vv, ny, t.out);
  t.cx[0] = nlk_tv_clamp((int)uu - sx, nx, t.out);
  t.cy[0] = nlk_tv_clamp((int)vv - sx, ny, t.out);
  t.cx[2] = nlk_tv_clamp((int)uu + sx, nx, t.out);
  t.cy[2] = nlk_tv_clamp((int)vv + sy, ny, t.out);
  t.cx[3] = nlk_tv_clamp((int)uu + 2 * sx, nx, t.out);
  t.cy[3] = nlk_tv_clamp((int)vv + 2 * sy, ny, t.out);
  t.tx = uu - t.cx[1];
  t.ty = vv - t.cy[1];
  return t;
}

__device__ __forceinline__ float nlk_tv_sample(const float* __restrict__ im, const NlkTvTaps& t, int nx) {
  double col[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double tap[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) tap[b] = im[t.cx[a] + nx * t.cy[b]];
    col[a] = nlk_tv_cubic(tap, t.ty);
  }
  return (float)nlk_tv_cubic(col, t.tx);
}

// ---- normalisation to 0..255 (reference: tvl1flow_lib.c:283-341)
__device__ __forceinline__ int nlk_tv_ord(float f) {  // order-preserving float -> int
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float nlk_tv_unord(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void k_tv_minmax(const float* __restrict__ a, const float* __restrict__ b, int n,
                            int* __restrict__ mm /* [0] = min, [1] = max, ordered ints */) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = a[i], y = b[i];
    lo = fminf(lo, fminf(x, y));
    hi = fmaxf(hi, fmaxf(x, y));
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], nlk_tv_ord(lo));
    atomicMax(&mm[1], nlk_tv_ord(hi));
  }
}

__global__ void k_tv_normalize(const float* __restrict__ a, const float* __restrict__ b,
                               float* __restrict__ oa, float* __restrict__ ob, int n,
                               const int* __restrict__ mm) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float lo = nlk_tv_unord(mm[0]), hi = nlk_tv_unord(mm[1]);
  const float den = hi - lo;
  oa[i] = den > 0 ? (float)(255.0 * (a[i] - lo) / den) : a[i];
  ob[i] = den > 0 ? (float)(255.0 * (b[i] - lo) / den) : b[i];
}

// ---- Gaussian, one direction per launch (reference: mask.c:221-330). g.b = the normalised
// half kernel (double, computed on the host exactly like the reference does, passed by value).
// The left pad mirrors about sample 0 without repeating it, the right pad repeats the last one.
struct NlkTvGauss {
  double b[32];
  int rad;
};

__global__ void k_tv_gauss(const float* __restrict__ in, float* __restrict__ out, int nx, int ny,
                           NlkTvGauss g, int vertical) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= nx || y >= ny) return;
  const int n = vertical ? ny : nx, c = vertical ? y : x;
  const int stride = vertical ? nx : 1;
  const float* line = in + (vertical ? x : y * nx);
  auto at = [&](int i) -> double {  // sample i of the padded line, i in [-rad, n + rad)
    const int k = i < 0 ? -i : (i >= n ? 2 * n - 1 - i : i);
    return (double)line[k * stride];
  };
  double sum = g.b[0] * at(c);
  for (int j = 1; j < g.rad; ++j) sum += g.b[j] * (at(c - j) + at(c + j));
  out[y * nx + x] = (float)sum;
}

__global__ void k_tv_reset(NlkTvState* st) {
  st->stop_iter = NLK_TV_MAXIT;
  st->iters = 0;
  st->error = INFINITY;
}

__global__ void k_tv_init_minmax(int* mm) {
  mm[0] = 0x7FFFFFFF;
  mm[1] = (int)0x80000000;
}

// ---- resampling (reference: zoom.c:44-108): out[i1][j1] = in(j1 / fx, i1 / fy) * gain
__global__ void k_tv_zoom(const float* __restrict__ in, float* __restrict__ out, int nx, int ny,
                          int nxx, int nyy, float fx, float fy, float gain, int use_gain) {
#pragma clang fp contract(off)
  const int j1 = blockIdx.x * blockDim.x + threadIdx.x, i1 = blockIdx.y * blockDim.y + threadIdx.y;
  if (j1 >= nxx || i1 >= nyy) return;
  const NlkTvTaps t = nlk_tv_taps((float)j1 / fx, (float)i1 / fy, nx, ny);
  const float g = nlk_tv_sample(in, t, nx);  // border_out = false
  out[i1 * nxx + j1] = use_gain ? g * gain : g;
}

// ---- centred gradient (reference: mask.c:148-214)
__global__ void k_tv_centered_grad(const float* __restrict__ f, float* __restrict__ dx,
                                   float* __restrict__ dy, int nx, int ny) {
#pragma clang fp contract(off)
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= nx || i >= ny) return;
  const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
  const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
  dx[i * nx + j] = (float)(0.5 * (f[i * nx + jr] - f[i * nx + jl]));
  dy[i * nx + j] = (float)(0.5 * (f[id * nx + j] - f[iu * nx + j]));
}

// ---- start of a warp: I1, I1x, I1y sampled at x + u (zero outside), |grad|^2 and the constant
// part of rho (reference: tvl1flow_lib.c:144-162)
__global__ void k_tv_warp(const float* __restrict__ I0, const float* __restrict__ I1,
                          const float* __restrict__ I1x, const float* __restrict__ I1y,
                          const float* __restrict__ u1, const float* __restrict__ u2,
                          float* __restrict__ I1wx, float* __restrict__ I1wy,
                          float* __restrict__ grad, float* __restrict__ rho_c, int nx, int ny) {
#pragma clang fp contract(off)
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= nx || i >= ny) return;
  const int p = i * nx + j;
  const float a = u1[p], b = u2[p];
  const NlkTvTaps t = nlk_tv_taps((float)(j + a), (float)(i + b), nx, ny);
  float w = 0.f, wx = 0.f, wy = 0.f;
  if (!t.out) {
    w = nlk_tv_sample(I1, t, nx);
    wx = nlk_tv_sample(I1x, t, nx);
    wy = nlk_tv_sample(I1y, t, nx);
  }
  const float Ix2 = wx * wx, Iy2 = wy * wy;
  I1wx[p] = wx;
  I1wy[p] = wy;
  grad[p] = Ix2 + Iy2;
  rho_c[p] = w - wx * a - wy * b - I0[p];
}

// ---- iteration n, first half: thresholding step, divergence of the dual variables, new flow,
// squared update (reference: tvl1flow_lib.c:172-230, mask.c:43-96). u is updated in place
// (read and written at the own pixel only). Each workgroup leaves its partial sum of the
// squared update in part[]; k_tv_dual adds them in a fixed order.
__device__ __forceinline__ float nlk_tv_div(const float* __restrict__ v1, const float* __restrict__ v2,
                                            int p, int i, int j, int nx, int ny) {
#pragma clang fp contract(off)
  const bool top = i == 0, bot = i == ny - 1, lef = j == 0, rig = j == nx - 1;
  if (!lef && !rig) {
    const float ax = v1[p] - v1[p - 1];
    if (!top && !bot) return ax + (v2[p] - v2[p - nx]);
    return top ? ax + v2[p] : ax - v2[p - nx];
  }
  if (!top && !bot) return lef ? v1[p] + v2[p] - v2[p - nx] : -v1[p - 1] + v2[p] - v2[p - nx];
  if (top) return lef ? v1[p] + v2[p] : -v1[p - 1] + v2[p];
  return lef ? v1[p] - v2[p - nx] : -v1[p - 1] - v2[p - nx];
}

__global__ void __launch_bounds__(256)
k_tv_primal(const float* __restrict__ rho_c, const float* __restrict__ I1wx,
            const float* __restrict__ I1wy, const float* __restrict__ grad,
            float* __restrict__ u1, float* __restrict__ u2, const float* __restrict__ p11,
            const float* __restrict__ p12, const float* __restrict__ p21,
            const float* __restrict__ p22, float* __restrict__ part, const NlkTvState* __restrict__ st,
            int n, int nx, int ny, float l_t, float theta) {
#pragma clang fp contract(off)
  if (n > st->stop_iter) return;
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), i = blockIdx.y * 8 + (threadIdx.x >> 5);
  float e = 0.f;
  if (j < nx && i < ny) {
    const int p = i * nx + j;
    const float a = u1[p], b = u2[p], gx = I1wx[p], gy = I1wy[p], g = grad[p];
    const float rho = rho_c[p] + (gx * a + gy * b);
    float d1, d2;
    if (rho < -l_t * g) {
      d1 = l_t * gx;
      d2 = l_t * gy;
    } else if (rho > l_t * g) {
      d1 = -l_t * gx;
      d2 = -l_t * gy;
    } else if (g < 1E-10) {
      d1 = d2 = 0;
    } else {
      const float fi = -rho / g;
      d1 = fi * gx;
      d2 = fi * gy;
    }
    const float v1 = a + d1, v2 = b + d2;
    const float na = v1 + theta * nlk_tv_div(p11, p12, p, i, j, nx, ny);
    const float nb = v2 + theta * nlk_tv_div(p21, p22, p, i, j, nx, ny);
    u1[p] = na;
    u2[p] = nb;
    e = (na - a) * (na - a) + (nb - b) * (nb - b);
  }
  // fixed-order workgroup sum
  __shared__ float red[4];
  for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- iteration n, second half: forward gradient of the new flow and dual update (reference:
// tvl1flow_lib.c:233-250, mask.c:98-141); hypot and 1 + taut*g are evaluated in double there.
// Workgroup 0 also closes the iteration: it adds the partial sums of k_tv_primal (fixed order,
// so a run is reproducible; the reference adds the pixels one by one in float, which rounds
// differently in the last bits) and, if the update is small enough, makes the iterations
// after n no-ops. The stop test is the reference's `error > epsilon^2 && n < MAX_ITERATIONS`.
__global__ void __launch_bounds__(256)
k_tv_dual(const float* __restrict__ u1, const float* __restrict__ u2, float* __restrict__ p11,
          float* __restrict__ p12, float* __restrict__ p21, float* __restrict__ p22,
          const float* __restrict__ part, int nparts, NlkTvState* __restrict__ st, int n, int nx,
          int ny, float taut, float eps2) {
#pragma clang fp contract(off)
  if (n > st->stop_iter) return;
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), i = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (j < nx && i < ny) {
    const int p = i * nx + j;
    const float a = u1[p], b = u2[p];
    const float ax = j < nx - 1 ? u1[p + 1] - a : 0.f, ay = i < ny - 1 ? u1[p + nx] - a : 0.f;
    const float bx = j < nx - 1 ? u2[p + 1] - b : 0.f, by = i < ny - 1 ? u2[p + nx] - b : 0.f;
    const float g1 = (float)hypot((double)ax, (double)ay);
    const float g2 = (float)hypot((double)bx, (double)by);
    const float ng1 = (float)(1.0 + (double)(taut * g1));
    const float ng2 = (float)(1.0 + (double)(taut * g2));
    p11[p] = (p11[p] + taut * ax) / ng1;
    p12[p] = (p12[p] + taut * ay) / ng1;
    p21[p] = (p21[p] + taut * bx) / ng2;
    p22[p] = (p22[p] + taut * by) / ng2;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    __shared__ double red[4];
    double s = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) s += (double)part[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float err = (float)((red[0] + red[1]) + (red[2] + red[3]));
      err /= (float)(nx * ny);
      st->iters = n;
      st->error = err;
      if (!(err > eps2)) st->stop_iter = n;  // (n == MAX is the host's loop bound)
    }
  }
}

// ---- small helpers of the boundary
// luminance of an interleaved colour image (what the reference's reader hands to the flow:
// lib/iio/iio.c:1048-1056, 3993-3996), double sum rounded once
__global__ void k_tv_gray(const float* __restrict__ im, float* __restrict__ g, int n, int ch) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* q = im + (size_t)i * ch;
  g[i] = ch >= 3 ? (float)(.299 * q[0] + .587 * q[1] + .114 * q[2]) : q[0];
}

__global__ void k_tv_interleave(const float* __restrict__ u1, const float* __restrict__ u2,
                                float* __restrict__ flow, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flow[2 * i] = u1[i];
  flow[2 * i + 1] = u2[i];
}

// occlusion mask of the pipelines: 255 where |backward-difference divergence| > th, replicated
// border (reference: scripts/nlkalman-seq.sh:70-73, plambda float stack)
__global__ void k_tv_occlusion(const float* __restrict__ flow, float* __restrict__ mask, int nx,
                               int ny, float th) {
#pragma clang fp contract(off)
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= nx || i >= ny) return;
  const int p = i * nx + j, pl = i * nx + (j > 0 ? j - 1 : 0), pu = (i > 0 ? i - 1 : 0) * nx + j;
  const float a = flow[2 * p] - flow[2 * pl], b = flow[2 * p + 1] - flow[2 * pu + 1];
  mask[p] = (fabsf(a + b) > th ? 1.0f : 0.0f) * 255.0f;
}
