// tu_group12.hip — launcher of k_group12 (12x12, lane = (channel, row)): the comparison variant
// NLK_GROUP12_ROWS=1 of the packed-lane kernel (k_groupp.h)
#include "k_group12.h"
#include "nlk_internal.h"

namespace {

template <int PSZ, int CH, bool SMO>
int launch_group_fast_t(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                        const float* prev, float* acc, const uint8_t* active) {
  NlkGTile tl{};
  // LDS tile halo = reach of the dominant kind of group; the rare spatial-branch
  // groups of a temporal frame that reach further fall back to HBM atomics
  tl.wmax = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
  // 4 x 1 targets per wavefront measured best for the temporal radius (profiles/README.md);
  // with a wide halo the tile would leave LDS room for ~1 wavefront per SIMD: 2 x 1 then, and for
  // the 12x12 kernel (8.0 ms against 8.8 with 4 x 1 at C3).
  // NLK_GTX/NLK_GTY override for experiments
  tl.tgx = getenv("NLK_GTX") ? atoi(getenv("NLK_GTX")) : ((PSZ == 8 && tl.wmax > 6) || PSZ == 12 ? 2 : 4);
  tl.tgy = getenv("NLK_GTY") ? atoi(getenv("NLK_GTY")) : 1;
  tl.ntx = (g.ngx + tl.tgx - 1) / tl.tgx;
  tl.nty = (g.ngy + tl.tgy - 1) / tl.tgy;
  // psz 8 runs its DCTs on the matrix cores (k_group8m.h); NLK_GROUP_DPP selects the
  // register/DPP kernel (k_group8.h) for comparison. psz 12 runs the packed-lane kernel
  // (k_group12p.h); NLK_GROUP12_ROWS selects the lane = (channel, row) kernel (k_group12.h)
  const bool mfma = false;
  const bool packed = false;
  const int rw_max = (tl.tgx - 1) * g.step + 2 * tl.wmax + g.psz;
  tl.rh_max = (tl.tgy - 1) * g.step + 2 * tl.wmax + g.psz;
  if (mfma) {
    // one aggregation access = 4x4 pixels of each plane: row stride = 4 and plane
    // stride = 16 (mod 32 banks) make the 64 lanes hit every bank twice
    tl.rwp = rw_max + ((4 - rw_max) % 32 + 32) % 32;
    tl.plane = tl.rwp * tl.rh_max;
    tl.plane += ((16 - tl.plane) % 32 + 32) % 32;
  } else {
    tl.rwp = rw_max | 1;
    tl.plane = tl.rwp * tl.rh_max;
  }
  // (+ the 12x12 kernels' transposition scratch: k_group12.h, k_group12p.h)
  const size_t lds = sizeof(float) * ((size_t)(CH + 1) * tl.plane + 4 + NLK_T12_FLOATS);
  if (lds > 160 * 1024) return fail(c, NLK_EUNSUP, "aggregation tile needs %zu bytes of LDS", lds);
  void (*kern)(const float*, const float*, const float*, const uint8_t*, NlkGeom, NlkGTile,
               const uint32_t*, const NlkTarget*, const uint32_t*, const uint8_t*, const float*,
               const float*, float*);
  kern = k_group12<CH, SMO>;
  HIPCHK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  const float* basis = (const float*)c->tabs.p;
  hipLaunchKernelGGL(kern, dim3(nlk_xcd_grid(tl.ntx * tl.nty)), dim3(64), lds, c->rv.stream, img, cur, prev,
                     (const uint8_t*)c->vmap.p, g, tl, (const uint32_t*)c->rv.topk,
                     (const NlkTarget*)c->rv.tinfo, (const uint32_t*)c->rv.gcoords,
                     active, basis, basis + PSZ * PSZ, acc);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}


}  // namespace

int nlk_launch_group12(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                       float* acc, const uint8_t* active) {
#define NLK_FAST(C)                                                                        \
  if (g.ch == C)                                                                           \
    return g.smoother ? launch_group_fast_t<12, C, true>(c, g, img, cur, prev, acc, active) \
                      : launch_group_fast_t<12, C, false>(c, g, img, cur, prev, acc, active);
  NLK_FAST(1) NLK_FAST(3)
#undef NLK_FAST
  return fail(c, NLK_EUNSUP, "%d channels not supported (1 or 3)", g.ch);
}

const float* nlk_basis12_table(void) { return &NLK_C12[0][0]; }
