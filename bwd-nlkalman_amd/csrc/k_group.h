// k_group.h — per-target group processing: gather the kept patches, 2-D DCT,
// Welford statistics, Kalman / Wiener (or RTS-smoother) gain, shrinkage of the
// group members, inverse DCT and weighted aggregation
// (reference: src/nlkalman.c:713-932 filter, :1603-1845 smoother).
//
// One wavefront per active target. Every lane owns the coefficients
// e = lane + 64*r of the ch*psz^2 coefficient vector for the whole call, so the
// running statistics, gains and shrinkage are lane-local register work; only
// the small separable DCTs cross lanes, through LDS with the psz x psz basis
// resident in LDS (no FFTW). Group members are re-transformed in a second pass
// once the gains are known, so no per-group coefficient store is needed and the
// group size is unbounded (the smoother's 105 slots at sigma=40 included).
#pragma once
#include "nlk_common.h"

template <int PSZ, int CH>
struct GroupShape {
  static constexpr int P2 = PSZ * PSZ;
  static constexpr int E = CH * P2;
  static constexpr int NR = (E + 63) / 64;
};

// Separable 2-D transform of NSET coefficient sets held one element per
// (lane, r). tab = DCT basis C (forward) or its transpose (inverse):
//   pass 1: T[c][j][i] = sum_k X[c][i][k] * tab[j][k]
//   pass 2: Y[c][i][j] = sum_k tab[i][k] * T[c][j][k]
template <int PSZ, int CH, int NSET>
__device__ inline void nlk_dct2d(const float* __restrict__ tab, float* __restrict__ X,
                                 float* __restrict__ T,
                                 float (&val)[NSET][GroupShape<PSZ, CH>::NR], int lane) {
  using S = GroupShape<PSZ, CH>;
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < S::NR; ++r) {
      const int e = lane + 64 * r;
      if (e < S::E) X[s * S::E + e] = val[s][r];
    }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < S::NR; ++r) {
      const int e = lane + 64 * r;
      if (e < S::E) {
        const int c = e / S::P2, rem = e % S::P2, i = rem / PSZ, j = rem % PSZ;
        const float* x = X + s * S::E + c * S::P2 + i * PSZ;
        const float* b = tab + j * PSZ;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < PSZ; ++k) acc = fmaf(x[k], b[k], acc);
        T[s * S::E + c * S::P2 + j * PSZ + i] = acc;
      }
    }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < S::NR; ++r) {
      const int e = lane + 64 * r;
      if (e < S::E) {
        const int c = e / S::P2, rem = e % S::P2, i = rem / PSZ, j = rem % PSZ;
        const float* b = tab + i * PSZ;
        const float* tt = T + s * S::E + c * S::P2 + j * PSZ;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < PSZ; ++k) acc = fmaf(b[k], tt[k], acc);
        val[s][r] = acc;
      }
    }
  __syncthreads();
}

__device__ inline float nlk_wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int PSZ, int CH, bool SMO>
__global__ void __launch_bounds__(64)
k_group(const float* __restrict__ img,   // matching / statistics image (planar)
        const float* __restrict__ cur,   // image whose patches are filtered
        const float* __restrict__ prev,  // previous output or nullptr
        const uint8_t* __restrict__ vmap, NlkGeom g,
        const uint32_t* __restrict__ topk, const NlkTarget* __restrict__ tinfo,
        const uint32_t* __restrict__ gcoords, const uint8_t* __restrict__ active,
        const float* __restrict__ basis,   // [psz][psz] orthonormal DCT-II
        const float* __restrict__ window,  // [psz][psz] aggregation window
        float* __restrict__ acc) {
  using S = GroupShape<PSZ, CH>;
  constexpr int NR = S::NR, E = S::E, P2 = S::P2;
  __shared__ __attribute__((aligned(16))) float lds[3 * P2 + 4 * E];
  float* Cm = lds;            // C[k][j]
  float* Ct = Cm + P2;        // C^T
  float* Wn = Ct + P2;        // window
  float* X = Wn + P2;         // [2][E]
  float* T = X + 2 * E;       // [2][E]

  const int t = blockIdx.x;
  if (!active[t]) return;
  const NlkTarget info = tinfo[t];
  if (info.nagg == 0) return;
  const int lane = threadIdx.x;
  for (int i = lane; i < P2; i += 64) {
    const float b = basis[i];
    Cm[i] = b;
    Ct[(i % PSZ) * PSZ + i / PSZ] = b;
    Wn[i] = window[i];
  }
  __syncthreads();

  const size_t npix = (size_t)g.w * g.h;
  // per-slot constant offsets
  int poff[NR];   // offset of the element inside a planar image, relative to the patch origin
  bool live[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = lane + 64 * r;
    live[r] = e < E;
    const int ee = live[r] ? e : 0;
    const int c = ee / P2, rem = ee % P2;
    poff[r] = c * (int)npix + (rem / PSZ) * g.w + rem % PSZ;
  }
  const int gy = t / g.ngx, gx = t - gy * g.ngx;
  const int px = gx * g.step, py = g.oy + gy * g.step;
  const bool prev_p = info.flags & 1;
  const int k = info.nsel;
  const float s2 = g.sigma2;

  float M0[NR], M0V[NR], V0[NR], V01[NR], M1[NR], V1[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) M0[r] = M0V[r] = V0[r] = V01[r] = M1[r] = V1[r] = 0.f;

  // ---------------- pass A: statistics over the k kept candidates
  int np0 = 0, np1 = 0;
  float val[2][NR], nxt[2][NR];
  bool vnext = false;
  auto gather = [&](int i, float (&dst)[2][NR], bool& v) {
    const uint32_t q = topk[(size_t)t * g.kmax + i];
    const int org = nlk_y(q) * g.w + nlk_x(q);
    v = prev_p && vmap[org];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      dst[0][r] = live[r] ? img[org + poff[r]] : 0.f;
      dst[1][r] = (live[r] && v) ? prev[org + poff[r]] : 0.f;
    }
  };
  if (k > 0) gather(0, nxt, vnext);
  for (int i = 0; i < k; ++i) {
    const bool v = vnext;
#pragma unroll
    for (int r = 0; r < NR; ++r) { val[0][r] = nxt[0][r]; val[1][r] = nxt[1][r]; }
    if (i + 1 < k) gather(i + 1, nxt, vnext);
    if (v) nlk_dct2d<PSZ, CH, 2>(Cm, X, T, val, lane);
    else {
      float (&one)[1][NR] = reinterpret_cast<float (&)[1][NR]>(val);
      nlk_dct2d<PSZ, CH, 1>(Cm, X, T, one, lane);
    }
    np1++;
    const float inp1 = 1.f / (float)np1;
    float inp0 = 0.f;
    bool in_group = false;
    if (v) {
      np0++;
      inp0 = 1.f / (float)np0;
      in_group = np0 <= g.ntagg;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const float a = val[0][r];
      const float d1 = a - M1[r];
      M1[r] += d1 * inp1;
      V1[r] += d1 * (a - M1[r]);
      if (v) {
        const float b = val[1][r];
        if (SMO) {  // reference: :1659-1667
          const float d0 = b - M0[r];
          M0[r] += d0 * inp0;
          V0[r] += d0 * (b - M0[r]);
        } else {    // reference: :769-783
          const float d0 = b - M0V[r];
          M0V[r] += d0 * inp0;
          V0[r] += d0 * (b - M0V[r]);
          if (in_group) M0[r] += (b - M0[r]) * inp0;
        }
        const float tt = b - a;
        V01[r] += tt * tt;
      }
    }
  }

  // ---------------- gains (reference: :799-811, :859-904; smoother :1683-1776)
  const int nagg = info.nagg;
  float gain[NR], mean[NR];
  float part = 0.f;
  {
    const float inp1 = np1 ? 1.f / (float)np1 : 0.f;
    const float inp0 = np0 ? 1.f / (float)np0 : 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const float v1 = V1[r] * inp1;
      const float v0 = np0 ? V0[r] * inp0 : V0[r];
      const float v01 = np0 ? V01[r] * inp0 : V01[r];
      float a, term;
      if (SMO) {
        a = v1 / (v1 + g.beta_t * v01);
        const float pv = v0 - g.beta_t * v01;
        term = (1 - a * a) * v1 + a * a * (pv > 0.f ? pv : 0.f);
        mean[r] = 0.f;
      } else if (np0 > 0) {
        const float d = v01 - (g.have_basic ? 0.f : s2);
        const float v = v0 + (0.f > d ? 0.f : d);
        a = v / (v + g.beta_t * s2);
        term = (1 - a * a) * v + a * a * s2;
        mean[r] = M0[r];
      } else {
        const float d = v1 - (g.have_basic ? 0.f : s2);
        const float v = 0.f > d ? 0.f : d;
        a = v / (v + g.beta_x * s2);
        term = a * v;
        mean[r] = M1[r];
      }
      gain[r] = a;
      if (live[r]) part += term;
    }
  }
  // the reference adds the same per-coefficient terms once per group member
  float vp = nlk_wave_sum(part) * (float)nagg;
  const bool passthrough = SMO && np0 == 0;  // reference: :1795-1804
  if (passthrough) vp = 0.f;
  const float wgt = 1.f / (vp > 1e-6f ? vp : 1e-6f);

  // ---------------- pass B: shrink, invert and aggregate the group members
  const float* src = g.have_basic ? cur : img;
  for (int n = 0; n < nagg; ++n) {
    const uint32_t q = gcoords[(size_t)t * g.gstride + n];
    const int org = nlk_y(q) * g.w + nlk_x(q);
    float out[NR];
    if (passthrough) {
#pragma unroll
      for (int r = 0; r < NR; ++r) out[r] = live[r] ? cur[org + poff[r]] : 0.f;
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        val[0][r] = live[r] ? src[org + poff[r]] : 0.f;
        val[1][r] = (SMO && live[r]) ? prev[org + poff[r]] : 0.f;
      }
      if (SMO) nlk_dct2d<PSZ, CH, 2>(Cm, X, T, val, lane);
      else {
        float (&one)[1][NR] = reinterpret_cast<float (&)[1][NR]>(val);
        nlk_dct2d<PSZ, CH, 1>(Cm, X, T, one, lane);
      }
      float y[1][NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float a = gain[r];
        y[0][r] = SMO ? (1 - a) * val[0][r] + a * val[1][r]
                      : a * val[0][r] + (1 - a) * mean[r];
      }
      nlk_dct2d<PSZ, CH, 1>(Ct, X, T, y, lane);
#pragma unroll
      for (int r = 0; r < NR; ++r) out[r] = y[0][r];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (live[r]) {
        const int e = lane + 64 * r;
        const float ww = wgt * Wn[e % P2];
        unsafeAtomicAdd(acc + org + poff[r], ww * out[r]);
        if (e < P2) unsafeAtomicAdd(acc + (size_t)CH * npix + org + poff[r], ww);
      }
    }
  }
}
