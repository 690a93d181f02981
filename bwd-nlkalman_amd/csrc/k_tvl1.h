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
// measure: see k_tv_level).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tv_hypot.h"

#define NLK_TV_MAXIT 300  // reference: tvl1flow_lib.c:24
#define NLK_TV_THREADS 1024  // workgroup of the level kernel: 16 wavefronts = 64 x 16 pixels per tile
#define NLK_TV_WAVES (NLK_TV_THREADS / 64)

// device-side state of the solver
struct NlkTvState {
  int iters;       // fixed-point iterations executed so far (all levels and warps)
  int stop_iter;   // multi-launch driver: iterations n > stop_iter of the current warp are no-ops
  int last;        // multi-launch driver: last executed iteration of the current warp
  float error;     // its mean squared update
  int redo;        // blocked driver: iterations of batch redo_n0 to recompute (it over-ran the stop)
  int redo_n0;
  int fin_stop;    // blocked driver: stop found by the group's closing launch (never read on the device)
};

// The host's copy of the state, in page-locked host memory the kernels write directly: the launch
// that closes a group of batches posts the state and then the group's sequence number (system
// scope); the host spins on the number instead of paying a copy + stream synchronisation per group.
struct NlkTvMail {
  NlkTvState st;
  unsigned seq;
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

// ---- sum of one value per lane over the wavefront, in every lane: the butterfly v += v[lane ^ off] for off = 32, 16,
// 8, 4, 2, 1 - the same pairs in the same order as a chain of __shfl_xor, hence the same roundings - without the six
// LDS-crossbar round trips (ds_bpermute) of that chain: row swaps for 32 and 16 (the sum of the two results is
// own + partner in every lane), a rotation by 8 inside the rows of 16, two bank-masked row shifts for 4, quad
// permutations for 2 and 1. A reduction costs ~80 cycles instead of ~400 (it sits between two barriers of every
// half iteration).
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t nlk_tv_dpp(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, 0xf, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t nlk_tv_xor8(uint32_t x) { return nlk_tv_dpp<0x128 /* row_ror:8 */>(x, x); }
__device__ __forceinline__ uint32_t nlk_tv_xor4(uint32_t x) {
  uint32_t t = nlk_tv_dpp<0x104 /* row_shl:4: lane i reads i + 4 */, 0x5>(x, x);  // lanes 0-3, 8-11 of a row
  return nlk_tv_dpp<0x114 /* row_shr:4: lane i reads i - 4 */, 0xa>(t, x);        // lanes 4-7, 12-15
}
__device__ __forceinline__ uint32_t nlk_tv_xor2(uint32_t x) { return nlk_tv_dpp<0x4e /* quad_perm 2 3 0 1 */>(x, x); }
__device__ __forceinline__ uint32_t nlk_tv_xor1(uint32_t x) { return nlk_tv_dpp<0xb1 /* quad_perm 1 0 3 2 */>(x, x); }

__device__ __forceinline__ float nlk_tv_wave_sum(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  v += __uint_as_float(nlk_tv_xor8(__float_as_uint(v)));
  v += __uint_as_float(nlk_tv_xor4(__float_as_uint(v)));
  v += __uint_as_float(nlk_tv_xor2(__float_as_uint(v)));
  v += __uint_as_float(nlk_tv_xor1(__float_as_uint(v)));
  return v;
}
__device__ __forceinline__ double nlk_tv_wave_sum(double v) {
  auto halves = [](double d, uint32_t& lo, uint32_t& hi) { const uint64_t u = (uint64_t)__double_as_longlong(d); lo = (uint32_t)u; hi = (uint32_t)(u >> 32); };
  auto whole = [](uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); };
  uint32_t lo, hi;
  halves(v, lo, hi);
  {
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = whole(a[0], b[0]) + whole(a[1], b[1]);
  }
  halves(v, lo, hi);
  {
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = whole(a[0], b[0]) + whole(a[1], b[1]);
  }
  halves(v, lo, hi); v += whole(nlk_tv_xor8(lo), nlk_tv_xor8(hi));
  halves(v, lo, hi); v += whole(nlk_tv_xor4(lo), nlk_tv_xor4(hi));
  halves(v, lo, hi); v += whole(nlk_tv_xor2(lo), nlk_tv_xor2(hi));
  halves(v, lo, hi); v += whole(nlk_tv_xor1(lo), nlk_tv_xor1(hi));
  return v;
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
  // one pair of atomics per workgroup (thousands of them on two addresses cost 0.1 ms)
  __shared__ float s_lo[16], s_hi[16];
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) { lo = fminf(lo, s_lo[k]); hi = fmaxf(hi, s_hi[k]); }
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

// (blockIdx.z picks one of two images: both frames / both flow components go through in one launch)
__global__ void k_tv_gauss(const float* __restrict__ in_a, float* __restrict__ out_a,
                           const float* __restrict__ in_b, float* __restrict__ out_b, int nx, int ny,
                           NlkTvGauss g, int vertical) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= nx || y >= ny) return;
  const float* __restrict__ in = blockIdx.z ? in_b : in_a;
  float* __restrict__ out = blockIdx.z ? out_b : out_a;
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

__global__ void k_tv_init_minmax(int* mm) {
  mm[0] = 0x7FFFFFFF;
  mm[1] = (int)0x80000000;
}

// ---- resampling (reference: zoom.c:44-108): out[i1][j1] = in(j1 / fx, i1 / fy) * gain
__global__ void k_tv_zoom(const float* __restrict__ in_a, float* __restrict__ out_a,
                          const float* __restrict__ in_b, float* __restrict__ out_b, int nx, int ny,
                          int nxx, int nyy, float fx, float fy, float gain, int use_gain) {
#pragma clang fp contract(off)
  const int j1 = blockIdx.x * blockDim.x + threadIdx.x, i1 = blockIdx.y * blockDim.y + threadIdx.y;
  if (j1 >= nxx || i1 >= nyy) return;
  const float* __restrict__ in = blockIdx.z ? in_b : in_a;
  float* __restrict__ out = blockIdx.z ? out_b : out_a;
  const NlkTvTaps t = nlk_tv_taps((float)j1 / fx, (float)i1 / fy, nx, ny);
  const float g = nlk_tv_sample(in, t, nx);  // border_out = false
  out[i1 * nxx + j1] = use_gain ? g * gain : g;
}

// ---- one pyramid level (reference: tvl1flow_lib.c:93-275): centred gradient of I1, then per
// warp the sampling of I1 / I1x / I1y at x + u and the iterations {thresholding + divergence +
// flow update; forward gradient + dual update} until the mean squared update drops below
// epsilon^2. The per-pixel steps are the device functions below (the reference's arithmetic,
// type for type; u and p are updated in place: each step writes only what it read at the own
// pixel). Two drivers use them:
//  * the smallest levels (up to NLK_TV_WG_PIXELS pixels) run ENTIRELY inside one 1024-thread workgroup
//    (k_tv_level_wg): all warps and iterations, phases separated by workgroup barriers, the stop
//    test evaluated in the kernel. An iteration costs ~1 us there instead of two launches of
//    ~5 us each, and nothing is read back;
//  * larger levels launch one kernel per half iteration (k_tv_primal / k_tv_dual); the stop test
//    also lives on the device (iterations after the converged one return at once) and the host
//    looks at the state only between batches of iterations.
// (A single persistent kernel with grid-wide barriers was measured too: on this multi-XCD part a
// device-scope release/acquire per phase costs ~10 us, more than the launches it replaces.)
struct NlkTvLevel {
  const float *I0, *I1;
  float *u1, *u2;
  float *I1x, *I1y, *I1wx, *I1wy, *grad, *rho_c, *p11, *p12, *p21, *p22;
  float* part;      // one partial sum of squared updates per workgroup
  NlkTvState* st;   // iteration count of the level (accumulated)
  NlkTvMail* mail;  // host-resident copy (see NlkTvMail)
  int nx, ny, nwarps;
  float l_t, theta, taut, eps2;
};

// ---- arithmetic cores, shared by every driver (identical rounding everywhere)
// backward-difference divergence; the association order differs between the body, the
// first/last column and the corners in the reference (mask.c:52-96) and is kept. v1l = v1 of
// the left neighbour, v2u = v2 of the upper one (ignored where the image has none)
__device__ __forceinline__ float nlk_tv_div_core(float v1c, float v1l, float v2c, float v2u, bool top,
                                                 bool bot, bool lef, bool rig) {
#pragma clang fp contract(off)
  if (!lef && !rig) {
    const float ax = v1c - v1l;
    if (!top && !bot) return ax + (v2c - v2u);
    return top ? ax + v2c : ax - v2u;
  }
  if (!top && !bot) return lef ? v1c + v2c - v2u : -v1l + v2c - v2u;
  if (top) return lef ? v1c + v2c : -v1l + v2c;
  return lef ? v1c - v2u : -v1l - v2u;
}

// thresholding step + flow update (reference: tvl1flow_lib.c:172-230)
__device__ __forceinline__ void nlk_tv_primal_core(float rho_c, float gx, float gy, float g, float a,
                                                   float b, float dv1, float dv2, float l_t, float theta,
                                                   float& na, float& nb) {
#pragma clang fp contract(off)
  const float rho = rho_c + (gx * a + gy * b);
  // the four cases of the thresholding as selects (all four occur inside most wavefronts, so
  // branches only serialise them): same values as the reference's if / else chain
  const float lg = l_t * g;
  const float fi = -rho / g;                       // (unused where g < 1e-10: may be inf / nan there)
  const float t1 = l_t * gx, t2 = l_t * gy;
  float d1 = fi * gx, d2 = fi * gy;
  const bool flat = g < 1E-10, hi = rho > lg, lo = rho < -lg;
  d1 = flat ? 0.f : d1;
  d2 = flat ? 0.f : d2;
  d1 = hi ? -t1 : d1;
  d2 = hi ? -t2 : d2;
  d1 = lo ? t1 : d1;
  d2 = lo ? t2 : d2;
  const float v1 = a + d1, v2 = b + d2;
  na = v1 + theta * dv1;
  nb = v2 + theta * dv2;
}

// dual update from the forward differences of the new flow (reference: tvl1flow_lib.c:233-250);
// hypot and 1 + taut*g are evaluated in double there. hypot() in double, rounded to float: the
// squares of floats are exact in double, so sqrt(x*x + y*y) carries two roundings of 2^-53 and
// gives the same float as the C library's hypot (no scaling is needed for flow gradients);
// nlk_tv_hypot returns that float without paying for the correctly rounded double root
__device__ __forceinline__ void nlk_tv_dual_core(float& p11, float& p12, float& p21, float& p22, float ax,
                                                 float ay, float bx, float by, float taut) {
#pragma clang fp contract(off)
  const float g1 = nlk_tv_hypot(ax, ay);  // == (float)sqrt((double)ax * ax + (double)ay * ay), tv_hypot.h
  const float g2 = nlk_tv_hypot(bx, by);
  const float ng1 = (float)(1.0 + (double)(taut * g1));
  const float ng2 = (float)(1.0 + (double)(taut * g2));
  p11 = (p11 + taut * ax) / ng1;
  p12 = (p12 + taut * ay) / ng1;
  p21 = (p21 + taut * bx) / ng2;
  p22 = (p22 + taut * by) / ng2;
}

__device__ __forceinline__ float nlk_tv_div(const float* __restrict__ v1, const float* __restrict__ v2,
                                            int p, int i, int j, int nx, int ny) {
  const bool top = i == 0, bot = i == ny - 1, lef = j == 0, rig = j == nx - 1;
  return nlk_tv_div_core(v1[p], lef ? 0.f : v1[p - 1], v2[p], top ? 0.f : v2[p - nx], top, bot, lef, rig);
}

// centred gradient (reference: mask.c:148-214) and the zero start of the dual variables (:137-141)
__device__ __forceinline__ void nlk_tv_px_init(const NlkTvLevel& L, int i, int j) {
#pragma clang fp contract(off)
  const int nx = L.nx, ny = L.ny, p = i * nx + j;
  const int jl = j > 0 ? j - 1 : 0, jr = j < nx - 1 ? j + 1 : nx - 1;
  const int iu = i > 0 ? i - 1 : 0, id = i < ny - 1 ? i + 1 : ny - 1;
  L.I1x[p] = (float)(0.5 * (L.I1[i * nx + jr] - L.I1[i * nx + jl]));
  L.I1y[p] = (float)(0.5 * (L.I1[id * nx + j] - L.I1[iu * nx + j]));
  L.p11[p] = L.p12[p] = L.p21[p] = L.p22[p] = 0.f;
}

// start of a warp: I1, I1x, I1y sampled at x + u (zero outside), |grad|^2 and the constant part
// of rho (reference: tvl1flow_lib.c:144-162)
__device__ __forceinline__ void nlk_tv_px_warp(const NlkTvLevel& L, int i, int j) {
#pragma clang fp contract(off)
  const int nx = L.nx, p = i * nx + j;
  const float a = L.u1[p], b = L.u2[p];
  const NlkTvTaps t = nlk_tv_taps((float)(j + a), (float)(i + b), nx, L.ny);
  float w = 0.f, wx = 0.f, wy = 0.f;
  if (!t.out) {
    w = nlk_tv_sample(L.I1, t, nx);
    wx = nlk_tv_sample(L.I1x, t, nx);
    wy = nlk_tv_sample(L.I1y, t, nx);
  }
  const float Ix2 = wx * wx, Iy2 = wy * wy;
  L.I1wx[p] = wx;
  L.I1wy[p] = wy;
  L.grad[p] = Ix2 + Iy2;
  L.rho_c[p] = w - wx * a - wy * b - L.I0[p];
}

// first half of an iteration at one pixel of the level's arrays; returns the squared update
__device__ __forceinline__ float nlk_tv_px_primal(const NlkTvLevel& L, int i, int j) {
#pragma clang fp contract(off)
  const int nx = L.nx, p = i * nx + j;
  const float a = L.u1[p], b = L.u2[p];
  float na, nb;
  nlk_tv_primal_core(L.rho_c[p], L.I1wx[p], L.I1wy[p], L.grad[p], a, b,
                     nlk_tv_div(L.p11, L.p12, p, i, j, nx, L.ny), nlk_tv_div(L.p21, L.p22, p, i, j, nx, L.ny),
                     L.l_t, L.theta, na, nb);
  L.u1[p] = na;
  L.u2[p] = nb;
  return (na - a) * (na - a) + (nb - b) * (nb - b);
}

// second half: forward gradient of the new flow (reference: mask.c:98-141) and dual update
__device__ __forceinline__ void nlk_tv_px_dual(const NlkTvLevel& L, int i, int j) {
#pragma clang fp contract(off)
  const int nx = L.nx, ny = L.ny, p = i * nx + j;
  const float a = L.u1[p], b = L.u2[p];
  const float ax = j < nx - 1 ? L.u1[p + 1] - a : 0.f, ay = i < ny - 1 ? L.u1[p + nx] - a : 0.f;
  const float bx = j < nx - 1 ? L.u2[p + 1] - b : 0.f, by = i < ny - 1 ? L.u2[p + nx] - b : 0.f;
  float p11 = L.p11[p], p12 = L.p12[p], p21 = L.p21[p], p22 = L.p22[p];
  nlk_tv_dual_core(p11, p12, p21, p22, ax, ay, bx, by, L.taut);
  L.p11[p] = p11;
  L.p12[p] = p12;
  L.p21[p] = p21;
  L.p22[p] = p22;
}

#define NLK_TV_WG_PIXELS 2100  // largest level solved by one workgroup (measured: 60 x 34 yes, 120 x 68 no)

// fixed-order sum of one float per thread over a workgroup of NLK_TV_THREADS
__device__ __forceinline__ float nlk_tv_block_sum(float e, float* redf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  e = nlk_tv_wave_sum(e);  // (== the chain e += __shfl_xor(e, off) for off = 32 .. 1)
  __syncthreads();  // (redf may still be read from the previous call)
  if (lane == 0) redf[wave] = e;
  __syncthreads();
  float bs = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) bs += redf[k];
  return bs;
}
// The same inside a loop of iterations that END with a barrier: with two buffers taken in turn the barrier in front
// of the write can go (buffer `it & 1` was last read two iterations ago, with a barrier in between), which leaves
// ONE barrier here - the one that also publishes what the first half of the iteration wrote.
__device__ __forceinline__ float nlk_tv_block_sum2(float e, float (*redf2)[32], int it) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* redf = redf2[it & 1];
  e = nlk_tv_wave_sum(e);
  if (lane == 0) redf[wave] = e;
  __syncthreads();
  float bs = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) bs += redf[k];
  return bs;
}

__global__ void __launch_bounds__(NLK_TV_THREADS) k_tv_level_wg(NlkTvLevel L) {
#pragma clang fp contract(off)
  // the flow and the dual variables of the whole level live in LDS, the per-warp constants of a
  // thread's pixels in registers: an iteration costs two LDS round trips instead of two trips to L2
  constexpr int SL = (NLK_TV_WG_PIXELS + NLK_TV_THREADS - 1) / NLK_TV_THREADS;
  __shared__ float redf2[2][32];
  __shared__ float s_u1[NLK_TV_WG_PIXELS], s_u2[NLK_TV_WG_PIXELS];
  __shared__ float s_p11[NLK_TV_WG_PIXELS], s_p12[NLK_TV_WG_PIXELS], s_p21[NLK_TV_WG_PIXELS], s_p22[NLK_TV_WG_PIXELS];
  const int nx = L.nx, ny = L.ny, npix = nx * ny;
  int pi[SL], pj[SL];
  bool on[SL];
#pragma unroll
  for (int m = 0; m < SL; ++m) {
    const int p = threadIdx.x + NLK_TV_THREADS * m;
    on[m] = p < npix;
    pi[m] = p / nx;
    pj[m] = p - pi[m] * nx;
    if (on[m]) {
      nlk_tv_px_init(L, pi[m], pj[m]);  // (centred gradient of I1 for the sampling; p = 0)
      s_u1[p] = L.u1[p];
      s_u2[p] = L.u2[p];
      s_p11[p] = s_p12[p] = s_p21[p] = s_p22[p] = 0.f;
    }
  }
  __syncthreads();
  int total = 0;
  for (int wi = 0; wi < L.nwarps; ++wi) {
    float rc[SL], gx[SL], gy[SL], gr[SL];
#pragma unroll
    for (int m = 0; m < SL; ++m) {
      const int p = threadIdx.x + NLK_TV_THREADS * m;
      rc[m] = gx[m] = gy[m] = gr[m] = 0.f;
      if (!on[m]) continue;
      if (wi > 0) {  // the warp samples at x + u: own pixel only
        L.u1[p] = s_u1[p];
        L.u2[p] = s_u2[p];
      }
      nlk_tv_px_warp(L, pi[m], pj[m]);
      rc[m] = L.rho_c[p]; gx[m] = L.I1wx[p]; gy[m] = L.I1wy[p]; gr[m] = L.grad[p];
    }
    float err = INFINITY;
    int n = 0;
    while (err > L.eps2 && n < NLK_TV_MAXIT) {  // reference: tvl1flow_lib.c:166
      ++n;
      float e = 0.f;
#pragma unroll
      for (int m = 0; m < SL; ++m) {
        if (!on[m]) continue;
        const int p = threadIdx.x + NLK_TV_THREADS * m, i = pi[m], j = pj[m];
        const bool top = i == 0, bot = i == ny - 1, lef = j == 0, rig = j == nx - 1;
        const int pl = lef ? p : p - 1, pu = top ? p : p - nx;
        const float a = s_u1[p], b = s_u2[p];
        float na, nb;
        nlk_tv_primal_core(rc[m], gx[m], gy[m], gr[m], a, b,
                           nlk_tv_div_core(s_p11[p], s_p11[pl], s_p12[p], s_p12[pu], top, bot, lef, rig),
                           nlk_tv_div_core(s_p21[p], s_p21[pl], s_p22[p], s_p22[pu], top, bot, lef, rig),
                           L.l_t, L.theta, na, nb);
        s_u1[p] = na;
        s_u2[p] = nb;
        e += (na - a) * (na - a) + (nb - b) * (nb - b);
      }
      // fixed-order sum (the reference adds the pixels one by one in float, which rounds
      // differently in the last bits); its barriers also separate the two halves
      err = nlk_tv_block_sum2(e, redf2, n);
      err /= (float)npix;
#pragma unroll
      for (int m = 0; m < SL; ++m) {
        if (!on[m]) continue;
        const int p = threadIdx.x + NLK_TV_THREADS * m, i = pi[m], j = pj[m];
        const int pr = j < nx - 1 ? p + 1 : p, pd = i < ny - 1 ? p + nx : p;
        const float a = s_u1[p], b = s_u2[p];
        const float ax = j < nx - 1 ? s_u1[pr] - a : 0.f, ay = i < ny - 1 ? s_u1[pd] - a : 0.f;
        const float bx = j < nx - 1 ? s_u2[pr] - b : 0.f, by = i < ny - 1 ? s_u2[pd] - b : 0.f;
        float p11 = s_p11[p], p12 = s_p12[p], p21 = s_p21[p], p22 = s_p22[p];
        nlk_tv_dual_core(p11, p12, p21, p22, ax, ay, bx, by, L.taut);
        s_p11[p] = p11;
        s_p12[p] = p12;
        s_p21[p] = p21;
        s_p22[p] = p22;
      }
      __syncthreads();
    }
    total += n;
  }
#pragma unroll
  for (int m = 0; m < SL; ++m) {
    const int p = threadIdx.x + NLK_TV_THREADS * m;
    if (on[m]) {
      L.u1[p] = s_u1[p];
      L.u2[p] = s_u2[p];
    }
  }
  if (threadIdx.x == 0) L.st->iters += total;
}

// ---- multi-launch driver: 64 x 4 pixels per workgroup (a wavefront = 64 consecutive pixels)
__global__ void __launch_bounds__(256) k_tv_init(NlkTvLevel L) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j < L.nx && i < L.ny) nlk_tv_px_init(L, i, j);
}

__global__ void __launch_bounds__(256) k_tv_warp(NlkTvLevel L) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j < L.nx && i < L.ny) nlk_tv_px_warp(L, i, j);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {  // a new fixed-point loop starts
    L.st->stop_iter = NLK_TV_MAXIT;
    L.st->last = 0;
    L.st->error = INFINITY;
    L.st->redo = 0;
    L.st->redo_n0 = -1;
    L.st->fin_stop = NLK_TV_MAXIT;
  }
}

__global__ void __launch_bounds__(256) k_tv_primal(NlkTvLevel L, int n) {
  if (n > L.st->stop_iter) return;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  float e = 0.f;
  if (j < L.nx && i < L.ny) e = nlk_tv_px_primal(L, i, j);
  __shared__ float redf[4];
  e = nlk_tv_block_sum(e, redf);
  if (threadIdx.x == 0) L.part[blockIdx.y * gridDim.x + blockIdx.x] = e;
}

// Workgroup 0 also closes the iteration: it adds the partial sums of k_tv_primal in a fixed
// order and, if the update is small enough, turns the iterations after n into no-ops (the
// other workgroups of THIS launch still run: they test n > stop_iter, and stop_iter >= n).
__global__ void __launch_bounds__(256) k_tv_dual(NlkTvLevel L, int n, int nparts) {
  if (n > L.st->stop_iter) return;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j < L.nx && i < L.ny) nlk_tv_px_dual(L, i, j);
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    __shared__ double red[4];
    double s = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) s += (double)L.part[k];
    s = nlk_tv_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float err = (float)((red[0] + red[1]) + (red[2] + red[3]));
      err /= (float)(L.nx * L.ny);
      L.st->last = n;
      L.st->error = err;
      L.st->iters += 1;
      if (!(err > L.eps2)) L.st->stop_iter = n;  // reference: tvl1flow_lib.c:166
    }
  }
}

// ---- blocked driver: NLK_TV_K iterations per launch (temporal blocking). On the levels that
// do not fit one workgroup a launch costs ~4 us whatever it does, and a full-size iteration
// moves 88 B per pixel through the caches; both go down by running K iterations on a tile
// held in LDS with a halo of K pixels (one iteration needs its neighbours' state one pixel
// further in every direction: p from the left / top for the divergence, u from the right /
// bottom for the gradient). Every region pixel is recomputed in every half iteration; what is
// computed from stale halo data never reaches the tile. The tile's pixels get exactly the
// values of the plain recursion.
//   k_tv_block  reads state `in` (u, p after n0 iterations), writes `out` (after n0 + count)
//               and one partial sum of squared updates per iteration and workgroup. The NEXT
//               launch starts by adding the partial sums in a fixed order (every workgroup does,
//               so that all take the same decision without a kernel in between): it finds the
//               first iteration that satisfies the stop test; later batches are no-ops, and if
//               the batch ran past the stop, that batch is redone from its input with fewer
//               iterations by the launch that closes a group of batches (mode 1), which also
//               judges the group's last batch. A launch never reads a state field it writes
//               (workgroups of one launch do not run together), except `stop_iter`, where both
//               values lead to the same action.
#define NLK_TV_K 4   // iterations per launch where the grid fills the chip (halo work grows with K),
#define NLK_TV_K2 8  // and on the small levels, whose launches are latency-bound
#define NLK_TV_TW 64
#define NLK_TV_TH 16   // tile rows: 16 (region 72 x 24: 2 pixel slots per thread for 1 tile pixel) or, for levels
#define NLK_TV_TH2 32  // with enough tiles to fill the chip, 32 (72 x 40: 3 slots for 2 tile pixels)
#define NLK_TV_BT 1024  // threads of a k_tv_block workgroup: the coarse levels have few tiles, so a tile
                        // must finish fast rather than leave room for others;
#ifndef NLK_TV_BT2
#define NLK_TV_BT2 512  // large grids: two workgroups per CU overlap one's loads / stores with the other's arithmetic
#endif

struct NlkTvBuf {
  float *u1, *u2, *p11, *p12, *p21, *p22;
};

// the stop test over the partial sums of one batch, by every thread of the workgroup alike:
// number of iterations of the batch that count (the first with error <= eps^2 is the last one)
template <int KI>
__device__ __forceinline__ int nlk_tv_judge(const NlkTvLevel& L, const float* __restrict__ part, int nblocks,
                                            int count, double (*red)[4], float* errs, bool& stop, float& err) {
  if (threadIdx.x < 256) {
    double s[KI];
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      s[k] = 0.0;
      if (k < count)
        for (int b = threadIdx.x; b < nblocks; b += 256) s[k] += (double)part[k * nblocks + b];
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      s[k] = nlk_tv_wave_sum(s[k]);
      if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < KI)
    errs[threadIdx.x] = (float)((red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3])) /
                        (float)(L.nx * L.ny);
  __syncthreads();
  int done = count;
  stop = false;
  for (int k = 0; k < count; ++k)
    if (!(errs[k] > L.eps2)) {  // reference: tvl1flow_lib.c:166
      done = k + 1;
      stop = true;
      break;
    }
  err = errs[done - 1];
  return done;
}

__device__ __forceinline__ void nlk_tv_post(const NlkTvLevel& L, unsigned seq) {
  NlkTvMail* m = L.mail;
  m->st = *L.st;  // (fields written by earlier launches or, just now, by this very thread)
  __threadfence_system();
  __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// mode 0: a batch; 2: a batch that first judges the previous one; 1: closes a group of batches
// (n0, count = the group's last batch, judged here; in / out = the buffers of the warp's FIRST
// batch); 3: closes a group whose batches were all judged by k_tv_decide.
// Judging inside the batches saves a launch per batch where the grid is small (every workgroup
// re-adds all partial sums: cheap for a few hundred workgroups, not for thousands)
template <int TW, int TH, int BT, int KI>
__global__ void __launch_bounds__(BT, 4)  // (>= 4 wavefronts per SIMD: 16 per CU either way)
k_tv_block(NlkTvLevel L, NlkTvBuf in, NlkTvBuf out, int n0, int count, int mode, unsigned seq) {
#pragma clang fp contract(off)
  constexpr int RW = TW + 2 * KI, RH = TH + 2 * KI, RPT = (RW * RH + BT - 1) / BT;
  __shared__ double red[KI][4];
  __shared__ float errs[KI];
  const int nblocks = gridDim.x * gridDim.y, block = blockIdx.y * gridDim.x + blockIdx.x;
  if (mode == 1 || mode == 3) {
    const bool poster = block == 0 && threadIdx.x == 0;
    if (L.st->stop_iter < NLK_TV_MAXIT) {  // found by a batch of this group
      if (poster) nlk_tv_post(L, seq);
      if (L.st->redo == 0) return;
      n0 = L.st->redo_n0;
      count = L.st->redo;
    } else if (mode == 3) {
      if (poster) nlk_tv_post(L, seq);
      return;
    } else {
      bool stop;
      float err;
      const int done = nlk_tv_judge<KI>(L, L.part + ((n0 / KI) & 1) * KI * nblocks, nblocks, count, red,
                                    errs, stop, err);
      if (block == 0 && threadIdx.x == 0) {
        L.st->iters += done;
        L.st->last = n0 + done;
        L.st->error = err;
        L.st->fin_stop = stop ? n0 + done : NLK_TV_MAXIT;
        nlk_tv_post(L, seq);
      }
      if (!stop || done == count) return;
      count = done;
    }
    if ((n0 / KI) & 1) { const NlkTvBuf t = in; in = out; out = t; }
  } else {
    if (n0 >= L.st->stop_iter) return;
    if (mode == 2) {
      const int n0p = n0 - KI;
      bool stop;
      float err;
      const int done = nlk_tv_judge<KI>(L, L.part + ((n0p / KI) & 1) * KI * nblocks, nblocks, KI,
                                    red, errs, stop, err);
      if (block == 0 && threadIdx.x == 0) {
        L.st->iters += done;
        L.st->last = n0p + done;
        L.st->error = err;
        if (stop) {
          if (done < KI) { L.st->redo = done; L.st->redo_n0 = n0p; }
          L.st->stop_iter = n0p + done;
        }
      }
      if (stop) return;
    }
  }
  float* const part = L.part + ((n0 / KI) & 1) * KI * nblocks;
  __shared__ float s_u1[RW * RH], s_u2[RW * RH];
  __shared__ float s_p11[RW * RH], s_p12[RW * RH];
  __shared__ float s_p21[RW * RH], s_p22[RW * RH];
  __shared__ float redf2[2][32];
  const int nx = L.nx, ny = L.ny;
  const int rx0 = blockIdx.x * TW - KI, ry0 = blockIdx.y * TH - KI;
  // the region pixels of this thread: index in the region, in the image, constants of the warp
  int gidx[RPT];
  bool on[RPT], mine[RPT];
  float rc[RPT], gx[RPT], gy[RPT], gr[RPT];
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = threadIdx.x + BT * m;
    const int ly = r / RW, lx = r - ly * RW;
    const int j = rx0 + lx, i = ry0 + ly;
    on[m] = r < RW * RH && j >= 0 && j < nx && i >= 0 && i < ny;
    mine[m] = on[m] && lx >= KI && lx < KI + TW && ly >= KI && ly < KI + TH;
    gidx[m] = on[m] ? i * nx + j : 0;
    rc[m] = L.rho_c[gidx[m]]; gx[m] = L.I1wx[gidx[m]]; gy[m] = L.I1wy[gidx[m]]; gr[m] = L.grad[gidx[m]];
    if (r < RW * RH) {
      s_u1[r] = on[m] ? in.u1[gidx[m]] : 0.f;
      s_u2[r] = on[m] ? in.u2[gidx[m]] : 0.f;
      s_p11[r] = on[m] ? in.p11[gidx[m]] : 0.f;
      s_p12[r] = on[m] ? in.p12[gidx[m]] : 0.f;
      s_p21[r] = on[m] ? in.p21[gidx[m]] : 0.f;
      s_p22[r] = on[m] ? in.p22[gidx[m]] : 0.f;
    }
  }
  __syncthreads();
  for (int k = 0; k < count; ++k) {
    float e = 0.f;
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      if (!on[m]) continue;
      const int r = threadIdx.x + BT * m;
      const int ly = r / RW, lx = r - ly * RW;
      const int j = rx0 + lx, i = ry0 + ly;
      const int rl = lx > 0 ? r - 1 : r, ru = ly > 0 ? r - RW : r;  // (region edge: value unused or stale halo)
      const float a = s_u1[r], b = s_u2[r];
      const float p11c = s_p11[r], p11l = s_p11[rl], p12c = s_p12[r], p12u = s_p12[ru];
      const float p21c = s_p21[r], p21l = s_p21[rl], p22c = s_p22[r], p22u = s_p22[ru];
      float dv1 = (p11c - p11l) + (p12c - p12u), dv2 = (p21c - p21l) + (p22c - p22u);  // interior pixel
      if (__builtin_expect(i == 0 || i == ny - 1 || j == 0 || j == nx - 1, 0)) {
        const bool top = i == 0, bot = i == ny - 1, lef = j == 0, rig = j == nx - 1;
        dv1 = nlk_tv_div_core(p11c, p11l, p12c, p12u, top, bot, lef, rig);
        dv2 = nlk_tv_div_core(p21c, p21l, p22c, p22u, top, bot, lef, rig);
      }
      float na, nb;
      nlk_tv_primal_core(rc[m], gx[m], gy[m], gr[m], a, b,
                         dv1, dv2, L.l_t, L.theta, na, nb);
      s_u1[r] = na;  // (u is read at the own pixel only in this half)
      s_u2[r] = nb;
      if (mine[m]) e += (na - a) * (na - a) + (nb - b) * (nb - b);
    }
    e = nlk_tv_block_sum2(e, redf2, k);  // (its barrier also publishes the new u)
    if (threadIdx.x == 0) part[k * nblocks + block] = e;
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      if (!on[m]) continue;
      const int r = threadIdx.x + BT * m;
      const int ly = r / RW, lx = r - ly * RW;
      const int j = rx0 + lx, i = ry0 + ly;
      const int rr = lx < RW - 1 ? r + 1 : r, rd = ly < RH - 1 ? r + RW : r;
      const float a = s_u1[r], b = s_u2[r];
      const float ax = j < nx - 1 ? s_u1[rr] - a : 0.f, ay = i < ny - 1 ? s_u1[rd] - a : 0.f;
      const float bx = j < nx - 1 ? s_u2[rr] - b : 0.f, by = i < ny - 1 ? s_u2[rd] - b : 0.f;
      float p11 = s_p11[r], p12 = s_p12[r], p21 = s_p21[r], p22 = s_p22[r];
      nlk_tv_dual_core(p11, p12, p21, p22, ax, ay, bx, by, L.taut);
      s_p11[r] = p11;  // (p is read at the own pixel only in this half)
      s_p12[r] = p12;
      s_p21[r] = p21;
      s_p22[r] = p22;
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    if (!mine[m]) continue;
    const int r = threadIdx.x + BT * m;
    out.u1[gidx[m]] = s_u1[r];
    out.u2[gidx[m]] = s_u2[r];
    out.p11[gidx[m]] = s_p11[r];
    out.p12[gidx[m]] = s_p12[r];
    out.p21[gidx[m]] = s_p21[r];
    out.p22[gidx[m]] = s_p22[r];
  }
}

// large grids: one workgroup judges batch (n0, count) between two batches
template <int KI>
__global__ void __launch_bounds__(256) k_tv_decide(NlkTvLevel L, int n0, int count, int nblocks) {
  if (n0 >= L.st->stop_iter) return;
  __shared__ double red[KI][4];
  __shared__ float errs[KI];
  bool stop;
  float err;
  const int done = nlk_tv_judge<KI>(L, L.part + ((n0 / KI) & 1) * KI * nblocks, nblocks, count, red, errs,
                                stop, err);
  if (threadIdx.x == 0) {
    L.st->iters += done;
    L.st->last = n0 + done;
    L.st->error = err;
    if (stop) {
      if (done < count) { L.st->redo = done; L.st->redo_n0 = n0; }
      L.st->stop_iter = n0 + done;
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
