// k_dct12.h — 12-point orthonormal DCT-II / DCT-III on 12 registers of one lane, as a
// flow graph instead of the 12x12 matrix product (the reference's FFTW REDFT10 / REDFT01 with its
// own scaling, src/nlkalman.c:204-220, 281-298, 335-353, is the same linear map).
//
//   y[k] = s_k sum_j x[j] cos(pi (2j+1) k / 24),  s_0 = sqrt(1/12), s_k = sqrt(1/6)
//
// Even/odd folding twice on the even half (12 -> 6 -> 3) and the shared pair (k = 3, 9) on the
// odd half: 66 operations forward, 68 inverse, against 84 for the folded matrix form and 144 for
// the plain one. Every step is an add, a multiply or an fma on independent registers, so the 64
// lanes of a wavefront run 64 transforms at once with no cross-lane traffic.
//
// Why this and not the matrix cores for 12x12 (measured rates: tools/ubench): a dense
// 12-point pass on v_mfma_f32_16x16x4_f32 fills 12 of the tile's 16 output columns, cannot use the
// even/odd symmetry (a block-diagonal operand costs what a dense one does) and so spends 4.5 MFMAs =
// 144 cycles per 12x12 plane at best (6 = 192 without an LDS repack between the passes); this graph
// costs 24 x 67 lane-operations = 27 wavefront instructions per plane at 60 active lanes, ~90 cycles.
#pragma once

#ifndef NLK_HD  // (a host test defines it as `static inline` and compiles this header with g++)
#include <hip/hip_runtime.h>
#define NLK_HD __device__ __forceinline__
#endif

namespace nlk_d12 {
// cos(N pi / 24) and cos(N pi / 12) folded with the orthonormal scale sqrt(1/6); S0 = sqrt(1/12)
constexpr float S0 = 0.288675134594812882f;
constexpr float S = 0.408248290463863016f;
constexpr float SC4 = 0.353553390593273762f;   // S * cos(pi/6)        (k = 4)
constexpr float K1 = 0.394337567297406441f;    // S * cos(pi/12)
constexpr float K3 = 0.288675134594812882f;    // S * cos(3 pi/12)
constexpr float K5 = 0.105662432702593559f;    // S * cos(5 pi/12)
constexpr float E1 = 0.404755671693680959f;    // S * cos(pi/24)
constexpr float E3 = 0.377172238691401321f;    // S * cos(3 pi/24)
constexpr float E5 = 0.323885156996537374f;    // S * cos(5 pi/24)
constexpr float E7 = 0.248525813115479607f;    // S * cos(7 pi/24)
constexpr float E9 = 0.156229852500726520f;    // S * cos(9 pi/24)
constexpr float E11 = 0.053287081695543585f;   // S * cos(11 pi/24)
}  // namespace nlk_d12

// forward: p[0..11] (samples) -> p[0..11] (coefficients)
NLK_HD void nlk_dct12_fast_fwd(float (&p)[12]) {
  using namespace nlk_d12;
  float s[6], d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { s[i] = p[i] + p[11 - i]; d[i] = p[i] - p[11 - i]; }
  // even half: 6-point DCT-II of s, folded again
  const float p0 = s[0] + s[5], p1 = s[1] + s[4], p2 = s[2] + s[3];
  const float q0 = s[0] - s[5], q1 = s[1] - s[4], q2 = s[2] - s[3];
  const float t = p0 + p2;
  p[0] = S0 * (t + p1);
  p[4] = SC4 * (p0 - p2);
  p[8] = S * __builtin_fmaf(0.5f, t, -p1);
  p[2] = __builtin_fmaf(K5, q2, __builtin_fmaf(K3, q1, K1 * q0));
  p[6] = K3 * (q0 - q1 - q2);
  p[10] = __builtin_fmaf(K1, q2, __builtin_fmaf(-K3, q1, K5 * q0));
  // odd half: k = 3 and 9 share two sums, the other four rows are dense
  const float P = d[0] - d[3] - d[4], Q = d[1] - d[2] - d[5];
  p[3] = __builtin_fmaf(E9, Q, E3 * P);
  p[9] = __builtin_fmaf(-E3, Q, E9 * P);
  p[1] = __builtin_fmaf(E11, d[5], __builtin_fmaf(E9, d[4], __builtin_fmaf(E7, d[3], __builtin_fmaf(E5, d[2], __builtin_fmaf(E3, d[1], E1 * d[0])))));
  p[5] = __builtin_fmaf(E7, d[5], __builtin_fmaf(E3, d[4], __builtin_fmaf(-E11, d[3], __builtin_fmaf(-E1, d[2], __builtin_fmaf(-E9, d[1], E5 * d[0])))));
  p[7] = __builtin_fmaf(-E5, d[5], __builtin_fmaf(-E9, d[4], __builtin_fmaf(E1, d[3], __builtin_fmaf(-E11, d[2], __builtin_fmaf(-E3, d[1], E7 * d[0])))));
  p[11] = __builtin_fmaf(-E1, d[5], __builtin_fmaf(E3, d[4], __builtin_fmaf(-E5, d[3], __builtin_fmaf(E7, d[2], __builtin_fmaf(-E9, d[1], E11 * d[0])))));
}

// inverse (DCT-III): y[0..11] (coefficients) -> y[0..11] (samples); the transposed graph
NLK_HD void nlk_dct12_fast_inv(float (&y)[12]) {
  using namespace nlk_d12;
  // even half
  const float a0 = S0 * y[0], a4 = SC4 * y[4], a8 = S * y[8];
  const float t = __builtin_fmaf(0.5f, a8, a0);
  const float ee0 = t + a4, ee1 = a0 - a8, ee2 = t - a4;
  const float eo0 = __builtin_fmaf(K5, y[10], __builtin_fmaf(K3, y[6], K1 * y[2]));
  const float eo1 = K3 * (y[2] - y[6] - y[10]);
  const float eo2 = __builtin_fmaf(K1, y[10], __builtin_fmaf(-K3, y[6], K5 * y[2]));
  float E[6], O[6];
  E[0] = ee0 + eo0; E[5] = ee0 - eo0;
  E[1] = ee1 + eo1; E[4] = ee1 - eo1;
  E[2] = ee2 + eo2; E[3] = ee2 - eo2;
  // odd half
  const float U = __builtin_fmaf(E9, y[9], E3 * y[3]), V = __builtin_fmaf(-E3, y[9], E9 * y[3]);
  O[0] = __builtin_fmaf(E11, y[11], __builtin_fmaf(E7, y[7], __builtin_fmaf(E5, y[5], E1 * y[1]))) + U;
  O[1] = __builtin_fmaf(-E9, y[11], __builtin_fmaf(-E3, y[7], __builtin_fmaf(-E9, y[5], E3 * y[1]))) + V;
  O[2] = __builtin_fmaf(E7, y[11], __builtin_fmaf(-E11, y[7], __builtin_fmaf(-E1, y[5], E5 * y[1]))) - V;
  O[3] = __builtin_fmaf(-E5, y[11], __builtin_fmaf(E1, y[7], __builtin_fmaf(-E11, y[5], E7 * y[1]))) - U;
  O[4] = __builtin_fmaf(E3, y[11], __builtin_fmaf(-E9, y[7], __builtin_fmaf(E3, y[5], E9 * y[1]))) - U;
  O[5] = __builtin_fmaf(-E1, y[11], __builtin_fmaf(-E5, y[7], __builtin_fmaf(E7, y[5], E11 * y[1]))) - V;
#pragma unroll
  for (int i = 0; i < 6; ++i) { y[i] = E[i] + O[i]; y[11 - i] = E[i] - O[i]; }
}
