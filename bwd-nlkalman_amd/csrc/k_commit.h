// k_commit.h — replay of the reference's raster-order "processed" mask
// (reference: src/nlkalman.c:597-600 skip test, :930-931 mark; smoother
// :1490-1493, :1843-1844).
//
// In the reference a target is skipped when an EARLIER, non-skipped target's
// group contained it. Whether a target is skipped therefore depends only on
// integer records (group coordinates, np0) that the matching kernel already
// produced for every target, not on any filtered pixel. A group reaches at most
// R = max(wsz)/step grid cells, so the serial order can be replayed as a
// wavefront: target (i, j) is decided at time i + (R+1)*j, when every target
// that can mark it has already been decided. One workgroup, one thread per
// grid row, one barrier per time step; the mask lives in LDS as one bit per
// grid target (only grid-aligned coordinates are ever tested).
#pragma once
#include "nlk_common.h"

__global__ void __launch_bounds__(1024)
k_mask_commit(const uint64_t* __restrict__ marks, uint8_t* __restrict__ active, int ngx,
              int ngy, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t* bits = (uint32_t*)smem;
  const int nwords = (ngx * ngy + 31) / 32;
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) bits[i] = 0;
  __syncthreads();
  const int side = 2 * R + 1;
  const int nsteps = ngx + (R + 1) * (ngy - 1);
  for (int s = 0; s < nsteps; ++s) {
    for (int j = threadIdx.x; j < ngy; j += blockDim.x) {
      const int i = s - (R + 1) * j;
      if (i < 0 || i >= ngx) continue;
      const int t = j * ngx + i;
      const int done = (bits[t >> 5] >> (t & 31)) & 1;
      active[t] = !done;
      if (done) continue;
      uint64_t m = marks[t];
      while (m) {
        const int b = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const int dj = b / side - R, di = b - (b / side) * side - R;
        const int jj = j + dj, ii = i + di;
        if (jj < 0 || jj >= ngy || ii < 0 || ii >= ngx) continue;
        const int tt = jj * ngx + ii;
        atomicOr(&bits[tt >> 5], 1u << (tt & 31));
      }
    }
    __syncthreads();
  }
}
