// tu_group12.hip — launcher of k_group12 (12x12, lane = (channel, row)): the comparison variant
// NLK_GROUP12_ROWS=1 of the packed-lane kernel (k_groupp.h)
#include "k_group12.h"
#include "nlk_internal.h"

namespace {

template <int CH, bool SMO>
int launch_group12_t(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                     const float* prev, float* acc, const uint8_t* active) {
  constexpr int PSZ = 12;
  if (c->deterministic)
    return fail(c, NLK_EUNSUP, "deterministic aggregation is not available in the NLK_GROUP12_ROWS variant");
  NlkGTile tl{};
  tl.wmax = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
  tl.tgx = getenv("NLK_GTX") ? atoi(getenv("NLK_GTX")) : 2;  // (8.0 ms against 8.8 with 4 x 1 at C3)
  tl.tgy = getenv("NLK_GTY") ? atoi(getenv("NLK_GTY")) : 1;
  tl.ntx = (g.ngx + tl.tgx - 1) / tl.tgx;
  tl.nty = (g.ngy + tl.tgy - 1) / tl.tgy;
  const int rw_max = (tl.tgx - 1) * g.step + 2 * tl.wmax + g.psz;
  tl.rh_max = (tl.tgy - 1) * g.step + 2 * tl.wmax + g.psz;
  tl.rwp = rw_max | 1;
  tl.plane = tl.rwp * tl.rh_max;
  // (+ the transposition scratch: k_group12.h)
  const size_t lds = sizeof(float) * ((size_t)(CH + 1) * tl.plane + 4 + NLK_T12_FLOATS);
  if (lds > 160 * 1024) return fail(c, NLK_EUNSUP, "aggregation tile needs %zu bytes of LDS", lds);
  auto kern = k_group12<CH, SMO>;
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
#define NLK_FAST(C)                                                                   \
  if (g.ch == C)                                                                      \
    return g.smoother ? launch_group12_t<C, true>(c, g, img, cur, prev, acc, active)  \
                      : launch_group12_t<C, false>(c, g, img, cur, prev, acc, active);
  NLK_FAST(1) NLK_FAST(3)
#undef NLK_FAST
  return fail(c, NLK_EUNSUP, "%d channels not supported (1 or 3)", g.ch);
}

const float* nlk_basis12_table(void) { return &NLK_C12[0][0]; }
