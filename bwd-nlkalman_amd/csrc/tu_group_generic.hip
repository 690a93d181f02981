// tu_group_generic.hip — launcher of the generic LDS-DCT group kernel (k_group.h), every patch size
#include "k_group.h"
#include "k_group_any.h"
#include "nlk_internal.h"

namespace {

template <int PSZ, int CH>
int launch_group_t(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                   const float* prev, float* acc, const uint8_t* active) {
  const int ngrid = g.ngx * g.ngy;
  const float* basis = (const float*)c->tabs.p;
  const float* window = basis + PSZ * PSZ;
  if (g.smoother)
    hipLaunchKernelGGL((k_group<PSZ, CH, true>), dim3(ngrid), dim3(64), 0, c->rv.stream, img,
                       cur, prev, (const uint8_t*)c->vmap.p, g, (const uint32_t*)c->rv.topk,
                       (const NlkTarget*)c->rv.tinfo, (const uint32_t*)c->rv.gcoords,
                       active, basis, window, acc);
  else
    hipLaunchKernelGGL((k_group<PSZ, CH, false>), dim3(ngrid), dim3(64), 0, c->rv.stream, img,
                       cur, prev, (const uint8_t*)c->vmap.p, g, (const uint32_t*)c->rv.topk,
                       (const NlkTarget*)c->rv.tinfo, (const uint32_t*)c->rv.gcoords,
                       active, basis, window, acc);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

template <int CH>
int launch_group_ch(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                    const float* prev, float* acc, const uint8_t* active) {
  switch (g.psz) {
    case 4: return launch_group_t<4, CH>(c, g, img, cur, prev, acc, active);
    case 6: return launch_group_t<6, CH>(c, g, img, cur, prev, acc, active);
    case 8: return launch_group_t<8, CH>(c, g, img, cur, prev, acc, active);
    case 10: return launch_group_t<10, CH>(c, g, img, cur, prev, acc, active);
    case 12: return launch_group_t<12, CH>(c, g, img, cur, prev, acc, active);
    case 16: return launch_group_t<16, CH>(c, g, img, cur, prev, acc, active);
  }
  return fail(c, NLK_EUNSUP, "patch size %d not supported (4, 6, 8, 10, 12, 16)", g.psz);
}


}  // namespace

// patch sizes 17..32, every channel count with ch * psz^2 <= 4096 (k_group_any.h)
int nlk_launch_group_any(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                         float* acc, const uint8_t* active) {
  if (c->deterministic)
    return fail(c, NLK_EUNSUP, "deterministic aggregation is not available for patch size %d with %d channels", g.psz, g.ch);
  if (g.psz > 32 || g.E > NLK_ANY_EMAX)
    return fail(c, NLK_EUNSUP, "patch size %d with %d channels not supported (patches up to 32 x 32, ch * psz^2 <= %d)",
                g.psz, g.ch, NLK_ANY_EMAX);
  const int ngrid = g.ngx * g.ngy;
  const float* basis = (const float*)c->tabs.p;
  const float* window = basis + g.p2;
  const size_t lds = sizeof(float) * (3 * (size_t)g.p2 + 4 * (size_t)g.E + NLK_ANY_NT / 64);
  void (*kern)(const float*, const float*, const float*, const uint8_t*, NlkGeom, const uint32_t*, const NlkTarget*,
               const uint32_t*, const uint8_t*, const float*, const float*, float*) =
      g.smoother ? k_group_any<true> : k_group_any<false>;
  HIPCHK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(ngrid), dim3(NLK_ANY_NT), lds, c->rv.stream, img, cur, prev, (const uint8_t*)c->vmap.p, g,
                     (const uint32_t*)c->rv.topk, (const NlkTarget*)c->rv.tinfo, (const uint32_t*)c->rv.gcoords, active,
                     basis, window, acc);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

int nlk_launch_group_generic(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                             const float* prev, float* acc, const uint8_t* active) {
  if (c->deterministic)
    return fail(c, NLK_EUNSUP, "deterministic aggregation is not available in the LDS-DCT kernel (candidate lists of "
                               "more than 128 entries / NLK_GENERIC_GROUP)");
  if (g.ch == 1) return launch_group_ch<1>(c, g, img, cur, prev, acc, active);
  if (g.ch == 3) return launch_group_ch<3>(c, g, img, cur, prev, acc, active);
  return fail(c, NLK_EUNSUP, "%d channels not supported (1 or 3)", g.ch);
}
