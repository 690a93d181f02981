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

static __global__ void __launch_bounds__(256)  // (static: the header is compiled into several translation units)
k_gather_tiles(float* __restrict__ acc, const float* __restrict__ slab, const uint8_t* __restrict__ tflag,
               NlkGeom g, NlkGTile tl, int nplanes) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= g.w) return;
  const int tpx = tl.tgx * g.step, tpy = tl.tgy * g.step;
  const int ex = (tl.tgx - 1) * g.step + tl.wmax + g.psz, ey = (tl.tgy - 1) * g.step + tl.wmax + g.psz;
  const int yy = y - g.oy;  // relative to the first target row
  const int tx_lo = max(0, (x - ex) / tpx), tx_hi = min(tl.ntx - 1, (x + tl.wmax) / tpx);
  const int ty_lo = max(0, yy - ey < 0 ? 0 : (yy - ey) / tpy), ty_hi = min(tl.nty - 1, (yy + tl.wmax) < 0 ? -1 : (yy + tl.wmax) / tpy);
  const size_t npix = (size_t)g.w * g.h, pix = (size_t)y * g.w + x;
  for (int p = 0; p < nplanes; ++p) {
    float s = 0.f;
    bool any = false;
    for (int ty = ty_lo; ty <= ty_hi; ++ty) {
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
        s += slab[(tile * nplanes + p) * tl.plane + (size_t)(y - ry0) * tl.rwp + (x - rx0)];
        any = true;
      }
    }
    if (any) acc[p * npix + pix] += s;
  }
}
