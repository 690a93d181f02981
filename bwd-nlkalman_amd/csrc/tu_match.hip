// tu_match.hip — launcher of the block-matching kernels (k_match.h)
#include "k_match.h"
#include "k_match_generic.h"
#include "nlk_internal.h"

namespace {

template <int PSZ, int CH, int MAXM>
int launch_match_t(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds,
                   const float* img, bool wide) {
  auto kern = wide ? k_bm_wide<PSZ, CH, MAXM> : k_bm_topk<PSZ, CH, MAXM>;
  if constexpr (PSZ >= 8 && (MAXM == 2 || (MAXM == 7 && PSZ == 8)))
    if (!wide && tl.bx == 2) kern = k_bm_topk<PSZ, CH, MAXM, 2>;
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

template <int PSZ, int CH>
int launch_match_m(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds,
                   const float* img, int maxm, bool wide) {
  if (maxm <= 2) return launch_match_t<PSZ, CH, 2>(c, g, tl, lds, img, wide);
  if (maxm <= 7) return launch_match_t<PSZ, CH, 7>(c, g, tl, lds, img, wide);
  return launch_match_t<PSZ, CH, 16>(c, g, tl, lds, img, wide);
}

template <int CH>
int launch_match_ch(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds,
                    const float* img, int maxm, bool wide) {
  switch (g.psz) {
    case 4: return launch_match_m<4, CH>(c, g, tl, lds, img, maxm, wide);
    case 6: return launch_match_m<6, CH>(c, g, tl, lds, img, maxm, wide);
    case 8: return launch_match_m<8, CH>(c, g, tl, lds, img, maxm, wide);
    case 10: return launch_match_m<10, CH>(c, g, tl, lds, img, maxm, wide);
    case 12: return launch_match_m<12, CH>(c, g, tl, lds, img, maxm, wide);
    case 16: return launch_match_m<16, CH>(c, g, tl, lds, img, maxm, wide);
  }
  return fail(c, NLK_EUNSUP, "patch size %d not supported (4, 6, 8, 10, 12, 16)", g.psz);
}

}  // namespace

int nlk_launch_match(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds, const float* img,
                     int maxm, bool wide) {
  if (g.ch == 1) return launch_match_ch<1>(c, g, tl, lds, img, maxm, wide);
  if (g.ch == 3) return launch_match_ch<3>(c, g, tl, lds, img, maxm, wide);
  return fail(c, NLK_EUNSUP, "%d channels not supported (1 or 3)", g.ch);
}


// the generic kernel: any patch size / channel count / search radius (k_match.h: k_bm_generic)
int nlk_launch_match_generic(nlk_ctx* c, const NlkGeom& g, const float* img) {
  const int wfull = 2 * max(g.wsz_x, g.wsz_t) + 1;
  const size_t lds = 8 * (size_t)g.kmax + 4 * (size_t)g.kmax + 4 * (size_t)g.gstride + 4 * (size_t)wfull * wfull + 16;
  if (lds > 160 * 1024)
    return fail(c, NLK_EUNSUP, "search window of %d candidates with k = %d needs %zu bytes of LDS (> 160 KiB)",
                wfull * wfull, g.kmax, lds);
  HIPCHK(c, hipFuncSetAttribute((const void*)k_bm_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_bm_generic, dim3(nlk_xcd_grid(g.ngx * g.ngy)), dim3(64), lds, c->rv.stream, img,
                     (const uint8_t*)c->vmap.p, g, g.kmax, c->rv.topk, c->rv.tinfo, c->rv.gcoords, c->rv.marks);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}
