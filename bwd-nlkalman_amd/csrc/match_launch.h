// match_launch.h — host-side launcher of k_bm_topk / k_bm_wide for one patch size (included by the
// tu_match_*.hip units, each of which instantiates a range of patch sizes so that `make -j` compiles them side
// by side: one unit with all six sizes took 3.5 minutes)
#pragma once
#include "k_match.h"
#include "nlk_internal.h"

namespace {


template <int PSZ, int CH, int MAXM, int ORD>
int launch_match_t(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds,
                   const float* img, bool wide) {
  auto kern = wide ? k_bm_wide<PSZ, CH, MAXM, ORD> : k_bm_topk<PSZ, CH, MAXM, 4, ORD>;
  if constexpr (PSZ >= 8 && (MAXM == 2 || (MAXM == 7 && PSZ == 8)))
    if (!wide && tl.bx == 2) kern = k_bm_topk<PSZ, CH, MAXM, 2, ORD>;
  HIPCHK(c, hipFuncSetAttribute((const void*)kern,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // k_bm_wide: the queue length is only known on the device, so a fixed grid strides over it
  const int grid = wide ? 512 : nlk_xcd_grid(tl.ntx * tl.nty);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(wide ? NLK_BM_THREADS : tl.threads), lds, c->rv.stream, img,
                     (const uint8_t*)c->vmap.p, g, tl, c->rv.topk,
                     c->rv.tinfo, c->rv.gcoords, c->rv.marks,
                     c->rv.wide + 1, c->rv.wide);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

template <int PSZ, int CH, int ORD>
int launch_match_m(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds,
                   const float* img, int maxm, bool wide) {
  if (maxm <= 2) return launch_match_t<PSZ, CH, 2, ORD>(c, g, tl, lds, img, wide);
  if (maxm <= 7) return launch_match_t<PSZ, CH, 7, ORD>(c, g, tl, lds, img, wide);
  return launch_match_t<PSZ, CH, 16, ORD>(c, g, tl, lds, img, wide);
}


template <int PSZ, int ORD = 0>
int launch_match_psz(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds, const float* img, int maxm, bool wide) {
  if (g.ch == 1) return launch_match_m<PSZ, 1, ORD>(c, g, tl, lds, img, maxm, wide);
  if (g.ch == 3) return launch_match_m<PSZ, 3, ORD>(c, g, tl, lds, img, maxm, wide);
  return fail(c, NLK_EUNSUP, "%d channels not supported (1 or 3)", g.ch);
}

}  // namespace

#define NLK_MATCH_PSZ(P)                                                                                      \
  int nlk_launch_match_p##P(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds, const float* img,    \
                            int maxm, bool wide) {                                                            \
    return launch_match_psz<P>(c, g, tl, lds, img, maxm, wide);                                               \
  }
// the same launcher for the opt-in block-summed distance order (k_match.h: nlk_match_block_sum; tu_match_f.hip)
#define NLK_MATCH_PSZ_BS(P)                                                                                   \
  int nlk_launch_match_p##P##_bs(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds, const float* img, \
                                 int maxm, bool wide) {                                                       \
    return launch_match_psz<P, 1>(c, g, tl, lds, img, maxm, wide);                                            \
  }
