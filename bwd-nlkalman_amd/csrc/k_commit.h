// k_commit.h — replay of the reference's raster-order "processed" mask
// (reference: src/nlkalman.c:597-600 skip test, :930-931 mark; smoother
// :1490-1493, :1843-1844).
//
// In the reference a target is skipped when an EARLIER, non-skipped target's
// group contained it. Whether a target is skipped therefore depends only on
// integer records (group coordinates, np0) that the matching kernel already
// produced for every target, not on any filtered pixel. A marking group
// reaches at most R grid cells, so the serial order can be replayed as a
// wavefront: target (i, j) is decided at time i + (R+1)*j, when every target
// that can mark it has already been decided. One workgroup, one thread per
// grid row (RPT rows when the grid has more than 1024 rows), one barrier per
// time step.
//
// The mask lives in LDS as one bit per grid target (only grid-aligned
// coordinates are ever tested). Only FORWARD marks (targets later in raster
// order) are applied: a bit is then only ever set by targets decided before
// its owner, so the final bit array IS the skip decision and nothing is
// written to HBM inside the loop.
#pragma once
#include "nlk_common.h"

template <int RPT>  // grid rows per thread: row j = threadIdx.x + r * blockDim.x
__global__ void __launch_bounds__(1024)
k_mask_commit(const uint64_t* __restrict__ marks, uint8_t* __restrict__ active, int ngx,
              int ngy, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t* bits = (uint32_t*)smem;
  const int ntot = ngx * ngy;
  // + a tail: marks aimed below the last grid row (strip mode) land in padding
  const int nwords = (ntot + (R + 1) * ngx + 63) / 32 + 1;
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) bits[i] = 0;
  __syncthreads();
  const int side = 2 * R + 1;
  const int skew = R + 1;
  const int nsteps = ngx + skew * (ngy - 1);
  auto fetch = [&](int j, int s) -> uint64_t {
    const int i = s - skew * j;
    return (j < ngy && i >= 0 && i < ngx) ? marks[(size_t)j * ngx + i] : 0ull;
  };
  // Decide the targets of time step s; mw = their mark words. Bit
  // (dj+R)*side + (di+R) of a mark word stands for the grid neighbour (di, dj);
  // the raster-forward neighbours are exactly the bits above the centre bit. Each
  // forward row of (up to) `side` neighbours is OR-ed into the bit array with at
  // most two 32-bit LDS atomics: no per-bit loop, no division.
  const int centre = R * side + R;
  const uint32_t rowmask = (1u << side) - 1u;
  // LDS atomics retire about one lane per clock on gfx950, so they are issued
  // only by the lanes that really have a bit to set.
  auto or_bits = [&](int pos, uint32_t mask) {  // bits[pos ...] |= mask
    const int wd = pos >> 5, sh = pos & 31;
    const uint64_t m2 = (uint64_t)mask << sh;
    if ((uint32_t)m2) atomicOr(&bits[wd], (uint32_t)m2);
    if ((uint32_t)(m2 >> 32)) atomicOr(&bits[wd + 1], (uint32_t)(m2 >> 32));
  };
  auto decide = [&](int s, const uint64_t (&mw)[RPT]) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int j = threadIdx.x + r * blockDim.x;
      const int i = s - skew * j;
      const bool in = j < ngy && i >= 0 && i < ngx;
      const int t = in ? j * ngx + i : 0;
      const bool done = (bits[t >> 5] >> (t & 31)) & 1u;
      const uint64_t fwd = (in && !done) ? (mw[r] >> (centre + 1)) : 0ull;
      or_bits(t + 1, (uint32_t)fwd & ((1u << R) - 1u));          // same row, di = 1..R
      for (int dj = 1; dj <= R; ++dj)                              // rows below, di = -R..R
        or_bits(t + dj * ngx - R, (uint32_t)(fwd >> (R + (dj - 1) * side)) & rowmask);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS atomics issued; loads stay in flight
    __builtin_amdgcn_s_barrier();
  };
  // The mark words of a row are consumed one per step. They are streamed in
  // phases of S steps: while phase p is decided from registers A, the loads of
  // phase p+1 into B are in flight, so no HBM latency sits between barriers.
  constexpr int S = 16;
  uint64_t A[RPT][S], B[RPT][S];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int e = 0; e < S; ++e) A[r][e] = fetch(threadIdx.x + r * blockDim.x, e);
  for (int s0 = 0; s0 < nsteps; s0 += S) {
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int e = 0; e < S; ++e) B[r][e] = fetch(threadIdx.x + r * blockDim.x, s0 + S + e);
#pragma unroll
    for (int e = 0; e < S; ++e) {
      uint64_t mw[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) mw[r] = A[r][e];
      decide(s0 + e, mw);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int e = 0; e < S; ++e) A[r][e] = B[r][e];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ntot; t += blockDim.x)
    active[t] = !((bits[t >> 5] >> (t & 31)) & 1u);
}
