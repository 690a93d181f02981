// tu_match.hip — dispatcher of the block-matching launchers (tu_match_a..e.hip, one range of patch sizes each)
// and launcher of the generic kernel (k_match_generic.h)
#include "k_match.h"
#include "k_match_generic.h"
#include "nlk_internal.h"

#define NLK_MATCH_DECL(P) \
  int nlk_launch_match_p##P(nlk_ctx*, const NlkGeom&, const NlkTile&, size_t, const float*, int, bool);
NLK_MATCH_DECL(4) NLK_MATCH_DECL(6) NLK_MATCH_DECL(8) NLK_MATCH_DECL(10) NLK_MATCH_DECL(12) NLK_MATCH_DECL(16)
int nlk_launch_match_p8_bs(nlk_ctx*, const NlkGeom&, const NlkTile&, size_t, const float*, int, bool);  // tu_match_f.hip

int nlk_launch_match(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds, const float* img,
                     int maxm, bool wide) {
  if (tl.order == 1 && g.psz == 8) return nlk_launch_match_p8_bs(c, g, tl, lds, img, maxm, wide);
  switch (g.psz) {
    case 4: return nlk_launch_match_p4(c, g, tl, lds, img, maxm, wide);
    case 6: return nlk_launch_match_p6(c, g, tl, lds, img, maxm, wide);
    case 8: return nlk_launch_match_p8(c, g, tl, lds, img, maxm, wide);
    case 10: return nlk_launch_match_p10(c, g, tl, lds, img, maxm, wide);
    case 12: return nlk_launch_match_p12(c, g, tl, lds, img, maxm, wide);
    case 16: return nlk_launch_match_p16(c, g, tl, lds, img, maxm, wide);
  }
  return fail(c, NLK_EUNSUP, "patch size %d not supported (4, 6, 8, 10, 12, 16)", g.psz);
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
