// k_dct8.h — 8-point orthonormal DCT-II / DCT-III on 8 registers of one lane as a flow graph
// (36 operations each way against 40 for the folded matrix form); companion of k_dct12.h.
//   y[k] = s_k sum_j x[j] cos(pi (2j+1) k / 16),  s_0 = sqrt(1/8), s_k = 1/2
#pragma once

#ifndef NLK_HD  // (a host test defines it as `static inline` and compiles this header with g++)
#include <hip/hip_runtime.h>
#define NLK_HD __device__ __forceinline__
#endif

namespace nlk_d8 {
constexpr float S0 = 0.353553390593273762f;   // sqrt(1/8) = cos(pi/4) / 2
constexpr float C1 = 0.461939766255643378f;   // cos(pi/8) / 2      (k = 2, 6)
constexpr float C3 = 0.191341716182544886f;   // cos(3 pi/8) / 2
constexpr float E1 = 0.490392640201615225f;   // cos(pi/16) / 2
constexpr float E3 = 0.415734806151272619f;   // cos(3 pi/16) / 2
constexpr float E5 = 0.277785116509801112f;   // cos(5 pi/16) / 2
constexpr float E7 = 0.097545161008064134f;   // cos(7 pi/16) / 2
}  // namespace nlk_d8

NLK_HD void nlk_dct8_fast_fwd(float (&p)[8]) {
  using namespace nlk_d8;
  float s[4], d[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { s[i] = p[i] + p[7 - i]; d[i] = p[i] - p[7 - i]; }
  const float p0 = s[0] + s[3], p1 = s[1] + s[2], q0 = s[0] - s[3], q1 = s[1] - s[2];
  p[0] = S0 * (p0 + p1);
  p[4] = S0 * (p0 - p1);
  p[2] = __builtin_fmaf(C3, q1, C1 * q0);
  p[6] = __builtin_fmaf(-C1, q1, C3 * q0);
  p[1] = __builtin_fmaf(E7, d[3], __builtin_fmaf(E5, d[2], __builtin_fmaf(E3, d[1], E1 * d[0])));
  p[3] = __builtin_fmaf(-E5, d[3], __builtin_fmaf(-E1, d[2], __builtin_fmaf(-E7, d[1], E3 * d[0])));
  p[5] = __builtin_fmaf(E3, d[3], __builtin_fmaf(E7, d[2], __builtin_fmaf(-E1, d[1], E5 * d[0])));
  p[7] = __builtin_fmaf(-E1, d[3], __builtin_fmaf(E3, d[2], __builtin_fmaf(-E5, d[1], E7 * d[0])));
}

NLK_HD void nlk_dct8_fast_inv(float (&y)[8]) {
  using namespace nlk_d8;
  const float a0 = S0 * y[0], a4 = S0 * y[4];
  const float ee0 = a0 + a4, ee1 = a0 - a4;
  const float eo0 = __builtin_fmaf(C3, y[6], C1 * y[2]), eo1 = __builtin_fmaf(-C1, y[6], C3 * y[2]);
  float E[4], O[4];
  E[0] = ee0 + eo0; E[3] = ee0 - eo0;
  E[1] = ee1 + eo1; E[2] = ee1 - eo1;
  O[0] = __builtin_fmaf(E7, y[7], __builtin_fmaf(E5, y[5], __builtin_fmaf(E3, y[3], E1 * y[1])));
  O[1] = __builtin_fmaf(-E5, y[7], __builtin_fmaf(-E1, y[5], __builtin_fmaf(-E7, y[3], E3 * y[1])));
  O[2] = __builtin_fmaf(E3, y[7], __builtin_fmaf(E7, y[5], __builtin_fmaf(-E1, y[3], E5 * y[1])));
  O[3] = __builtin_fmaf(-E1, y[7], __builtin_fmaf(E3, y[5], __builtin_fmaf(-E5, y[3], E7 * y[1])));
#pragma unroll
  for (int i = 0; i < 4; ++i) { y[i] = E[i] + O[i]; y[7 - i] = E[i] - O[i]; }
}
