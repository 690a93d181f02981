// groupp_launch.h — host-side launcher of k_groupp<PSZ, SMO> (included by the tu_groupp_*.hip units,
// each of which instantiates a range of patch sizes so that `make -j` compiles them side by side)
#pragma once
#include "k_gather.h"
#include "k_groupp.h"
#include "nlk_internal.h"

template <int PSZ>
static int nlk_groupp_launch_t(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                               const float* prev, float* acc, const uint8_t* active) {
  typedef NlkPP<PSZ> K;
  // Deterministic mode runs a temporal frame's far-reaching (spatial-branch) groups in a second
  // launch whose tiles have the spatial halo, so that no member ever leaves its tile (k_group8.h)
  const bool split = c->deterministic && g.have_prev && !g.smoother && g.wsz_x > g.wsz_t;
  const size_t ntiles = (size_t)g.ngx * g.ngy;
  size_t slab_off = 0;
  int* cnt_base = nullptr;
  for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
    NlkGTile tl{};
    tl.split = split;
    tl.far = pass;
    // LDS tile halo = reach of the dominant kind of group; without the split the rare spatial-branch
    // groups of a temporal frame that reach further go to HBM atomics
    tl.wmax = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
    if (pass == 1) tl.wmax = g.wsz_x;
    tl.tgx = tl.tgy = 1;  // one target per workgroup
    tl.ntx = g.ngx;
    tl.nty = g.ngy;
    const int rw_max = 2 * tl.wmax + g.psz;
    tl.rh_max = rw_max;
    // one aggregation access = PSZ rows x NBK blocks of PB pixels (lane = NBK * row + block): the row
    // stride with the fewest bank collisions among those addresses (two halves of 32 lanes)
    int best = 1 << 30;
    tl.rwp = rw_max;
    for (int r = rw_max; r < rw_max + 8; ++r) {
      int cost = 0;
      for (int half = 0; half < 2; ++half) {
        int cnt[32] = {0}, mx = 0;
        for (int l = 32 * half; l < 32 * half + 32 && l < PSZ * K::NBK; ++l) {
          const int v = ++cnt[((l / K::NBK) * r + K::PB * (l % K::NBK)) & 31];
          mx = v > mx ? v : mx;
        }
        cost += mx;
      }
      cost = cost * 64 + (r - rw_max);  // (ties: the narrowest)
      if (cost < best) { best = cost; tl.rwp = r; }
    }
    tl.plane = (tl.rwp * tl.rh_max + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)2 * tl.plane + K::SCRATCH + K::gains(g.ch));
    if (lds > 160 * 1024) return fail(c, NLK_EUNSUP, "aggregation tile needs %zu bytes of LDS", lds);
    if (c->deterministic) {
      if (pass == 0) {  // (both passes' slabs are sized before the first launch: growing the buffer frees it)
        size_t need = ntiles * (g.ch + 1) * tl.plane;
        if (split) need += ntiles * (g.ch + 1) * ((size_t)(2 * g.wsz_x + g.psz + 8) * (2 * g.wsz_x + g.psz) + 4);
        int rc;
        // (flags of both passes, then their tile-row counters: 2 x (1 + nty) ints, cleared per call)
        const size_t cnt_off = (2 * ntiles + 15) & ~(size_t)15, cnt_bytes = sizeof(int) * 2 * (1 + (size_t)g.ngy);
        if ((rc = reserve(c, c->slab, sizeof(float) * need)) || (rc = reserve(c, c->tflag, cnt_off + cnt_bytes))) return rc;
        HIPCHK(c, hipMemsetAsync((uint8_t*)c->tflag.p + cnt_off, 0, cnt_bytes, c->rv.stream));
        cnt_base = (int*)((uint8_t*)c->tflag.p + cnt_off);
      }
      tl.slab = (float*)c->slab.p + slab_off;
      tl.tflag = (uint8_t*)c->tflag.p + (size_t)pass * ntiles;
      tl.tcount = cnt_base + pass * (1 + g.ngy);
      slab_off += ntiles * (g.ch + 1) * tl.plane;
    }
    auto kern = g.smoother ? k_groupp<PSZ, true> : k_groupp<PSZ, false>;
    HIPCHK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float* basis = (const float*)c->tabs.p;
    hipLaunchKernelGGL(kern, dim3(nlk_xcd_grid(g.ngx * g.ngy)), dim3(64), lds, c->rv.stream, img, cur, prev, g, tl,
                       (const uint32_t*)c->rv.topk, (const NlkTarget*)c->rv.tinfo, (const uint32_t*)c->rv.gcoords,
                       active, basis, basis + PSZ * PSZ, acc);
    if (c->deterministic)
      nlk_launch_gather(c->rv.stream, acc, tl.slab, tl.tflag, tl.tcount, g, tl, g.ch + 1);
    HIPCHK(c, hipGetLastError());
  }
  return NLK_OK;
}
