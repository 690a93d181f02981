// (float)sqrt(x*x + y*y) evaluated in double, as the reference's dual update does
// (lib/tvl1flow/tvl1flow_lib.c:241-242: hypot() of two floats, assigned to a float), at a third
// of the cost of the correctly rounded double square root:
//   * x*x + y*y: the squares of floats are exact in double, the sum is rounded once (2^-53);
//   * g ~ sqrt(s) from v_rsq_f64 and ONE coupled Newton step (relative error <= 2^-47.7 over the
//     1.7e10 inputs of tools/ubench/hypot_check.hip, which also found no mismatch);
//   * g rounds to the same float as the correctly rounded root unless it lies within the error
//     bound of the midpoint of two floats (low 29 mantissa bits ~ 0x10000000): only then - one
//     input in 65 000, one wavefront in 1 000 - the exact root is computed.
// The result therefore equals (float)sqrt(s) for every input, which is what keeps the flow
// bit-identical with the reference.
#pragma once
#include <hip/hip_runtime.h>

// half width of the guard band around a float midpoint, in units of the double's last place:
// 2^12 ulp = 2^-40 relative, 200x the largest error the check program has seen
#define NLK_TV_HYPOT_BAND 0x1000u

__device__ __forceinline__ double nlk_tv_sqrt_fast(double s) {
#pragma clang fp contract(off)
  const double y = __builtin_amdgcn_rsq(s);
  const double g = s * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  return __builtin_fma(g, r, g);
}

__device__ __forceinline__ float nlk_tv_hypot(float x, float y) {
#pragma clang fp contract(off)
  const double dx = x, dy = y;
  const double s = dx * dx + dy * dy;
  const double g = nlk_tv_sqrt_fast(s);
  float gf = (float)g;
  const unsigned lo = (unsigned)__double_as_longlong(g) & 0x1fffffffu;
  const bool near = lo - (0x10000000u - NLK_TV_HYPOT_BAND) < 2u * NLK_TV_HYPOT_BAND;
  if (__builtin_expect(near, 0)) gf = (float)sqrt(s);
  return s == 0.0 ? 0.f : gf;  // (rsq(0) is infinite)
}
