// k_group_any.h — per-target group processing for the patch sizes the specialised kernels are not built for:
// 17..32, any parity, any channel count with ch * psz^2 <= 4096 (reference: src/nlkalman.c:524-525, 555-560 take
// any). Patch size and channel count are run-time values here.
//
// One workgroup of 256 threads per active target; thread e owns the coefficients e, e + 256, ... of the
// ch * psz^2 vector (at most 16 each: ch * psz^2 <= 4096), so the Welford statistics (reference: :769-783,
// smoother :1659-1667), gains and shrinkage are register work; the separable DCTs go through LDS with the basis
// resident there. Group members are re-transformed once the gains are known. Slow next to k_groupp / k_group8m
// (no data reuse, one target per workgroup, atomics to HBM for every member pixel): a completeness path.
#pragma once
#include "nlk_common.h"

#define NLK_ANY_NT 256
#define NLK_ANY_NR 16
#define NLK_ANY_EMAX (NLK_ANY_NT * NLK_ANY_NR)

// Separable 2-D transform of NSET coefficient sets held one element per (thread, r); tab = basis (forward) or
// its transpose (inverse). Same passes as nlk_dct2d (k_group.h).
template <int NSET>
__device__ inline void nlk_any_dct2d(const float* __restrict__ tab, float* __restrict__ X, float* __restrict__ T,
                                     float (&val)[NSET][NLK_ANY_NR], int tid, int psz, int p2, int E,
                                     const int (&ec)[NLK_ANY_NR], const int (&ei)[NLK_ANY_NR], const int (&ej)[NLK_ANY_NR]) {
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < NLK_ANY_NR; ++r) {
      const int e = tid + NLK_ANY_NT * r;
      if (e < E) X[s * E + e] = val[s][r];
    }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < NLK_ANY_NR; ++r) {
      const int e = tid + NLK_ANY_NT * r;
      if (e < E) {
        const float* x = X + s * E + ec[r] * p2 + ei[r] * psz;
        const float* b = tab + ej[r] * psz;
        float acc = 0.f;
        for (int k = 0; k < psz; ++k) acc = fmaf(x[k], b[k], acc);
        T[s * E + ec[r] * p2 + ej[r] * psz + ei[r]] = acc;
      }
    }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < NLK_ANY_NR; ++r) {
      const int e = tid + NLK_ANY_NT * r;
      if (e < E) {
        const float* b = tab + ei[r] * psz;
        const float* tt = T + s * E + ec[r] * p2 + ej[r] * psz;
        float acc = 0.f;
        for (int k = 0; k < psz; ++k) acc = fmaf(b[k], tt[k], acc);
        val[s][r] = acc;
      }
    }
  __syncthreads();
}

template <bool SMO>
__global__ void __launch_bounds__(NLK_ANY_NT)
k_group_any(const float* __restrict__ img,   // matching / statistics image (planar)
            const float* __restrict__ cur,   // image whose patches are filtered
            const float* __restrict__ prev,  // previous output or nullptr
            const uint8_t* __restrict__ vmap, NlkGeom g, const uint32_t* __restrict__ topk,
            const NlkTarget* __restrict__ tinfo, const uint32_t* __restrict__ gcoords,
            const uint8_t* __restrict__ active,
            const float* __restrict__ basis,   // [psz][psz] orthonormal DCT-II
            const float* __restrict__ window,  // [psz][psz] aggregation window
            float* __restrict__ acc) {
  constexpr int NR = NLK_ANY_NR, NT = NLK_ANY_NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [3 p2 + 4 E + NT / 64]
  const int psz = g.psz, p2 = g.p2, E = g.E, CH = g.ch;
  float* Cm = lds;        // C[k][j]
  float* Ct = Cm + p2;    // C^T
  float* Wn = Ct + p2;    // window
  float* X = Wn + p2;     // [2][E]
  float* T = X + 2 * E;   // [2][E]
  float* red = T + 2 * E; // [NT / 64]

  const int t = blockIdx.x;
  if (!active[t]) return;
  const NlkTarget info = tinfo[t];
  if (info.nagg == 0) return;
  const int tid = threadIdx.x;
  for (int i = tid; i < p2; i += NT) {
    const float b = basis[i];
    Cm[i] = b;
    Ct[(i % psz) * psz + i / psz] = b;
    Wn[i] = window[i];
  }
  __syncthreads();

  const size_t npix = (size_t)g.w * g.h;
  int poff[NR], ec[NR], ei[NR], ej[NR];
  bool live[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = tid + NT * r;
    live[r] = e < E;
    const int ee = live[r] ? e : 0;
    ec[r] = ee / p2;
    const int rem = ee - ec[r] * p2;
    ei[r] = rem / psz;
    ej[r] = rem - ei[r] * psz;
    poff[r] = ec[r] * (int)npix + ei[r] * g.w + ej[r];
  }
  const bool prev_p = info.flags & 1;
  const int k = info.nsel;
  const float s2 = g.sigma2;

  float M0[NR], M0V[NR], V0[NR], V01[NR], M1[NR], V1[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) M0[r] = M0V[r] = V0[r] = V01[r] = M1[r] = V1[r] = 0.f;

  // ---------------- pass A: statistics over the k kept candidates
  int np0 = 0, np1 = 0;
  float val[2][NR];
  for (int i = 0; i < k; ++i) {
    const uint32_t q = topk[(size_t)t * g.kmax + i];
    const int org = nlk_y(q) * g.w + nlk_x(q);
    const bool v = prev_p && vmap[org];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      val[0][r] = live[r] ? img[org + poff[r]] : 0.f;
      val[1][r] = (live[r] && v) ? prev[org + poff[r]] : 0.f;
    }
    if (v) nlk_any_dct2d<2>(Cm, X, T, val, tid, psz, p2, E, ec, ei, ej);
    else {
      float (&one)[1][NR] = reinterpret_cast<float (&)[1][NR]>(val);
      nlk_any_dct2d<1>(Cm, X, T, one, tid, psz, p2, E, ec, ei, ej);
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
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  float vp = 0.f;
  for (int i = 0; i < NT / 64; ++i) vp += red[i];
  vp *= (float)nagg;
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
      if (SMO) nlk_any_dct2d<2>(Cm, X, T, val, tid, psz, p2, E, ec, ei, ej);
      else {
        float (&one)[1][NR] = reinterpret_cast<float (&)[1][NR]>(val);
        nlk_any_dct2d<1>(Cm, X, T, one, tid, psz, p2, E, ec, ei, ej);
      }
      float y[1][NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float a = gain[r];
        y[0][r] = SMO ? (1 - a) * val[0][r] + a * val[1][r] : a * val[0][r] + (1 - a) * mean[r];
      }
      nlk_any_dct2d<1>(Ct, X, T, y, tid, psz, p2, E, ec, ei, ej);
#pragma unroll
      for (int r = 0; r < NR; ++r) out[r] = y[0][r];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (live[r]) {
        const float ww = wgt * Wn[ei[r] * psz + ej[r]];
        unsafeAtomicAdd(acc + org + poff[r], ww * out[r]);
        if (ec[r] == 0) unsafeAtomicAdd(acc + (size_t)CH * npix + org + poff[r], ww);
      }
    }
  }
}
