// k_match_generic.h — block matching for any patch size / channel count / search radius
// (included by tu_match.hip only: the kernel is not a template)
#pragma once
#include "k_match.h"

// ---------------------------------------------------------------------------
// Any patch size, channel count and search radius: one wavefront per target, the images read
// straight from HBM / L2 (no LDS window), distances kept as keys in LDS, the same selection
// (k smallest under the (distance, window index) order) by a bitwise radix select over those
// keys. Same element order and roundings as nlk_match_target, so the records are the oracle's
// bit for bit. Slow (no data reuse); it serves the parameter combinations the tiled kernels are
// not instantiated for: odd patch sizes, other channel counts than 1 / 3, windows of more than
// 1024 candidates (reference: src/nlkalman.c:524-525, 555-560, 637-639 accept them all).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_bm_generic(const float* __restrict__ img, const uint8_t* __restrict__ vmap, NlkGeom g, int ksel_max,
             uint32_t* __restrict__ topk, NlkTarget* __restrict__ tinfo, uint32_t* __restrict__ gcoords,
             uint64_t* __restrict__ marks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  const int ngrid = g.ngx * g.ngy;
  const int ti = nlk_xcd_tile(blockIdx.x, ngrid);
  if (ti >= ngrid) return;
  const size_t t = (size_t)ti;
  const int gy = ti / g.ngx, gx = ti - gy * g.ngx;
  const int psz = g.psz, px = gx * g.step, py = g.oy + gy * g.step;
  const int prev_p = g.have_prev ? vmap[(size_t)py * g.w + px] : 0;
  int k = prev_p ? g.npt : g.npx;
  NlkTarget info = {0, 0, 0, prev_p, {0ull, 0ull}};
  if (k <= 1) {  // single-patch mode (see k_bm_topk)
    if (g.smoother) {
      info.nagg = 1;
      if (lane == 0) gcoords[t * g.gstride] = nlk_pack_xy(px, py);
    }
    if (lane == 0) { tinfo[t] = info; marks[t] = 0; }
    return;
  }
  const int wsz = (g.smoother || prev_p) ? g.wsz_t : g.wsz_x;
  const int x0 = max(px - wsz, 0), x1 = min(px + wsz, g.w - psz) + 1;
  const int y0 = max(py - wsz, 0), y1 = min(py + wsz, g.h - psz) + 1;
  const int nwx = x1 - x0, n = nwx * (y1 - y0);
  k = min(k, n);
  const int wfull = 2 * max(g.wsz_x, g.wsz_t) + 1;
  uint64_t* surv = (uint64_t*)smem;                        // [ksel_max]
  uint32_t* sel = (uint32_t*)(surv + ksel_max);            // [ksel_max]
  uint32_t* grp = sel + ksel_max;                          // [gstride]
  uint32_t* keys = grp + g.gstride;                        // [wfull * wfull]
  (void)wfull;
  const size_t npix = (size_t)g.w * g.h;
  const float norm = (float)(psz * psz * g.ch);
  for (int i = lane; i < n; i += 64) {
    const int wy = i / nwx, wx = i - wy * nwx;
    const float* cp = img + (size_t)(y0 + wy) * g.w + x0 + wx;
    const float* tp = img + (size_t)py * g.w + px;
    float acc = 0.f;
    for (int hy = 0; hy < psz; ++hy)
      for (int hx = 0; hx < psz; ++hx)
        for (int c = 0; c < g.ch; ++c) {
#pragma clang fp contract(off)
          const float e = cp[c * npix + (size_t)hy * g.w + hx] - tp[c * npix + (size_t)hy * g.w + hx];
          const float e2 = e * e;
          acc = acc + e2;
        }
    const float q = acc / norm;  // IEEE division (reference: src/nlkalman.c:701)
    keys[i] = __float_as_uint(q > 0.f ? q : 0.f);
  }
  nlk_wave_lds_fence();
  // k-th smallest key: the bits from the top (keys are >= +0: unsigned order)
  uint32_t prefix = 0, pmask = 0;
  int kk = k;
  for (int b = 31; b >= 0; --b) {
    int cnt0 = 0;
    for (int i = lane; i < n; i += 64) {
      const uint32_t key = keys[i];
      cnt0 += ((key & pmask) == prefix) && !((key >> b) & 1u);
    }
    for (int off = 32; off > 0; off >>= 1) cnt0 += __shfl_xor(cnt0, off, 64);
    if (kk > cnt0) { prefix |= 1u << b; kk -= cnt0; }
    pmask |= 1u << b;
  }
  // survivors: every key below the k-th, and the first kk (in window order) equal to it
  int neq = 0, npos = 0;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const uint32_t key = i < n ? keys[i] : 0xffffffffu;
    const bool less = i < n && key < prefix, eq = i < n && key == prefix;
    const uint64_t be = __ballot(eq);
    const bool keep = less || (eq && neq + __popcll(be & lt_mask) < kk);
    const uint64_t bk = __ballot(keep);
    if (keep) surv[npos + __popcll(bk & lt_mask)] = ((uint64_t)key << 32) | (uint32_t)i;
    neq += __popcll(be);
    npos += __popcll(bk);
  }
  nlk_wave_lds_fence();
  for (int base = 0; base < k; base += 64) {
    const int p = base + lane;
    const uint64_t mine = p < k ? surv[p] : ~0ull;
    int rank = 0;
    for (int j = 0; j < k; ++j) rank += surv[j] < mine;
    if (p < k) {
      const int mi = (int)(uint32_t)mine;
      const int wy = mi / nwx, wx = mi - wy * nwx;
      sel[rank] = nlk_pack_xy(x0 + wx, y0 + wy);
    }
  }
  nlk_wave_lds_fence();
  nlk_match_epilogue(g, t, px, py, prev_p, k, sel, grp, vmap, topk, tinfo, gcoords, marks, lane);
}
