// k_gather.h — deterministic aggregation (NLK_DETERMINISTIC / nlk_ctx_set_deterministic).
//
// The group kernels normally add their private accumulator tiles to the frame accumulator with
// global float atomics, whose order differs from run to run (reference: the `omp atomic` adds of
// src/nlkalman.c:923-931 have the same property). In deterministic mode every workgroup instead
// WRITES its tile (all planes, as laid out in LDS) to a slab of its own, and this kernel adds, for
// every pixel, the slabs of the tiles that cover it in ascending tile order: one thread per pixel,
// no atomics, the same sum order every time.
#pragma once
#include "k_group8.h"  // NlkGTile
#include "nlk_common.h"

// `tcount[0]` = number of flagged tiles of the launch, `tcount[1 + ty]` = of tile row ty (added by the
// group kernels): a pixel skips tile rows, or the whole launch (the far pass of a frame without holes),
// that wrote nothing, without looking at their flags.
template <int NP>  // planes summed together (NP = ch + 1 for ch <= 3; more channels: one plane per pass)
static __global__ void __launch_bounds__(256)
k_gather_tiles(float* __restrict__ acc, const float* __restrict__ slab, const uint8_t* __restrict__ tflag,
               const int* __restrict__ tcount, NlkGeom g, NlkGTile tl, int nplanes, int p0) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= g.w || tcount[0] == 0) return;
  const int tpx = tl.tgx * g.step, tpy = tl.tgy * g.step;
  const int ex = (tl.tgx - 1) * g.step + tl.wmax + g.psz, ey = (tl.tgy - 1) * g.step + tl.wmax + g.psz;
  const int yy = y - g.oy;  // relative to the first target row
  const int tx_lo = max(0, (x - ex) / tpx), tx_hi = min(tl.ntx - 1, (x + tl.wmax) / tpx);
  const int ty_lo = yy - ey < 0 ? 0 : (yy - ey) / tpy, ty_hi = (yy + tl.wmax) < 0 ? -1 : min(tl.nty - 1, (yy + tl.wmax) / tpy);
  const size_t npix = (size_t)g.w * g.h, pix = (size_t)y * g.w + x;
  float s[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) s[p] = 0.f;
  bool any = false;
  for (int ty = ty_lo; ty <= ty_hi; ++ty) {
    if (tcount[1 + ty] == 0) continue;
    const int gy0 = ty * tl.tgy, cy = min(tl.tgy, g.ngy - gy0);
    const int ry0 = max(g.oy + gy0 * g.step - tl.wmax, 0);
    const int ry1 = min(g.oy + (gy0 + cy - 1) * g.step + tl.wmax + g.psz, g.h);
    if (y < ry0 || y >= ry1) continue;
    for (int tx = tx_lo; tx <= tx_hi; ++tx) {
      const int gx0 = tx * tl.tgx, cx = min(tl.tgx, g.ngx - gx0);
      const int rx0 = max(gx0 * g.step - tl.wmax, 0);
      const int rx1 = min((gx0 + cx - 1) * g.step + tl.wmax + g.psz, g.w);
      if (x < rx0 || x >= rx1) continue;
      const size_t tile = (size_t)ty * tl.ntx + tx;
      if (!tflag[tile]) continue;
      const float* sp = slab + (tile * nplanes + p0) * tl.plane + (size_t)(y - ry0) * tl.rwp + (x - rx0);
#pragma unroll
      for (int p = 0; p < NP; ++p) s[p] += sp[(size_t)p * tl.plane];
      any = true;
    }
  }
  if (any) {
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[(p0 + p) * npix + pix] += s[p];
  }
}

// host side: one launch for ch + 1 <= 4 planes, otherwise a launch per plane
static inline void nlk_launch_gather(hipStream_t stream, float* acc, const float* slab, const uint8_t* tflag,
                                     const int* tcount, const NlkGeom& g, const NlkGTile& tl, int nplanes) {
  const dim3 grid((g.w + 255) / 256, g.h), block(256);
  switch (nplanes) {
    case 2: hipLaunchKernelGGL(k_gather_tiles<2>, grid, block, 0, stream, acc, slab, tflag, tcount, g, tl, nplanes, 0); break;
    case 3: hipLaunchKernelGGL(k_gather_tiles<3>, grid, block, 0, stream, acc, slab, tflag, tcount, g, tl, nplanes, 0); break;
    case 4: hipLaunchKernelGGL(k_gather_tiles<4>, grid, block, 0, stream, acc, slab, tflag, tcount, g, tl, nplanes, 0); break;
    default:
      for (int p = 0; p < nplanes; ++p)
        hipLaunchKernelGGL(k_gather_tiles<1>, grid, block, 0, stream, acc, slab, tflag, tcount, g, tl, nplanes, p);
  }
}
