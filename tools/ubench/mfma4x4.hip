// Micro-benchmark + operand-layout check of v_mfma_f32_4x4x1_16B_f32 on gfx950: sixteen
// independent 4x4 outer-product accumulations per instruction, one block per lane quad.
//   hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip && ./mfma4x4
// Prints
//   1. the operand layout, checked against a scalar product (hypothesis: block = lane >> 2; A row = lane & 3;
//      B column = lane & 3; D[i][j] in register i of lane 4 * block + j), and whether a k = 1 step is one fused
//      multiply-add (one rounding) or a product rounded before the add;
//   2. the separable folded 8x8 DCT-II built from it (two stages of four k = 1 steps per parity quadrant, the
//      first stage's result used as the second stage's B operand WITHOUT a shuffle; the inverse with the data as
//      the A operand of its first stage), checked against a double-precision DCT of random patches;
//   3. issue rate (independent accumulators), dependent-accumulator latency, latency when a result is the next
//      instruction's B operand, and the price of v_fma_f32 fillers between the MFMAs, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- 1. layout
__global__ void k_layout(const float* a, const float* b, const float* c, float* d) {
  const int l = threadIdx.x;
  f4 C = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
  C = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], C, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[l * 4 + v] = C[v];
}

// ---------------------------------------------------------------- 2. separable folded DCT
// lane = 4 * patch + i: rows i and 7 - i of the patch (R[0..7], R[8..15]); E[par][k] = C[2 * (lane & 3) + par][k]
// (forward operand of both stages), G[par][a] = C[2 * a + par][lane & 3] (inverse operand of both stages)
__device__ __forceinline__ void fold(const float (&R)[16], float (&F)[4][4]) {
  float P[8], M[8];
  for (int c = 0; c < 8; ++c) { P[c] = R[c] + R[8 + c]; M[c] = R[c] - R[8 + c]; }
  for (int s = 0; s < 4; ++s) {
    F[0][s] = P[s] + P[7 - s]; F[1][s] = P[s] - P[7 - s];
    F[2][s] = M[s] + M[7 - s]; F[3][s] = M[s] - M[7 - s];
  }
}
__device__ __forceinline__ void unfold(const float (&F)[4][4], float (&R)[16]) {
  for (int s = 0; s < 4; ++s) {
    const float e0 = F[0][s] + F[1][s], e1 = F[0][s] - F[1][s];   // P[s], P[7-s]
    const float o0 = F[2][s] + F[3][s], o1 = F[2][s] - F[3][s];   // M[s], M[7-s]
    R[s] = e0 + o0; R[7 - s] = e1 + o1; R[8 + s] = e0 - o0; R[8 + 7 - s] = e1 - o1;
  }
}
// Y[q][a] = coefficient (2a + qr, 2 * (lane & 3) + qc) of the lane's patch
__device__ __forceinline__ void dct_fwd(const float (&F)[4][4], const float (&E)[2][4], f4 (&Y)[4]) {
  f4 T[4];
  for (int q = 0; q < 4; ++q) {
    T[q] = f4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) T[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(F[q][k], E[q & 1][k], T[q], 0, 0, 0);
  }
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 4; ++i) Y[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(E[q >> 1][i], T[q][i], Y[q], 0, 0, 0);
}
// X[q][m] = folded pixel (lane & 3, m) of quadrant q
__device__ __forceinline__ void dct_inv(const f4 (&Y)[4], const float (&G)[2][4], f4 (&X)[4]) {
  f4 U[4];
  for (int q = 0; q < 4; ++q) {
    U[q] = f4{0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a) U[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(Y[q][a], G[q >> 1][a], U[q], 0, 0, 0);
  }
  for (int q = 0; q < 4; ++q)
    for (int b = 0; b < 4; ++b) X[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(G[q & 1][b], U[q][b], X[q], 0, 0, 0);
}
__global__ void k_dct(const float* patches /* [16][8][8] */, const float* basis /* [8][8] */, float* coef /* [16][8][8] */,
                      float* back /* [16][8][8] */) {
  const int l = threadIdx.x, p = l >> 2, i = l & 3;
  float E[2][4], G[2][4];
  for (int par = 0; par < 2; ++par)
    for (int k = 0; k < 4; ++k) { E[par][k] = basis[(2 * i + par) * 8 + k]; G[par][k] = basis[(2 * k + par) * 8 + i]; }
  float R[16], F[4][4];
  for (int c = 0; c < 8; ++c) { R[c] = patches[p * 64 + i * 8 + c]; R[8 + c] = patches[p * 64 + (7 - i) * 8 + c]; }
  fold(R, F);
  f4 Y[4];
  for (int q = 0; q < 4; ++q) Y[q] = f4{0.f, 0.f, 0.f, 0.f};
  dct_fwd(F, E, Y);
  for (int q = 0; q < 4; ++q)
    for (int a = 0; a < 4; ++a) coef[p * 64 + (2 * a + (q >> 1)) * 8 + 2 * i + (q & 1)] = Y[q][a];
  f4 X[4];
  for (int q = 0; q < 4; ++q) X[q] = f4{0.f, 0.f, 0.f, 0.f};
  dct_inv(Y, G, X);
  float FF[4][4];
  // (the folds carry no factor: pixel (i, m) = X0 + X1 + X2 + X3, its mirrors with the signs of the parities)
  for (int q = 0; q < 4; ++q) for (int m = 0; m < 4; ++m) FF[q][m] = X[q][m];
  float O[16];
  unfold(FF, O);
  for (int c = 0; c < 8; ++c) { back[p * 64 + i * 8 + c] = O[c]; back[p * 64 + (7 - i) * 8 + c] = O[8 + c]; }
}

// ---------------------------------------------------------------- 3. rates
#define N_ITER 2000
// MODE 0: 8 independent accumulators; 1: one dependent accumulator chain; 2: each result is the next one's B
// operand (a chain through the B operand); 3: the 32-instruction forward transform of (2) back to back
template <int MODE, int NV>
__global__ void __launch_bounds__(64) k_rate(float* out, float a) {
  f4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  const float fa = threadIdx.x * 0.001f, fb = a;
  float E[2][4], F[4][4];
  for (int i = 0; i < 8; ++i) E[i >> 2][i & 3] = fb + i;
  for (int i = 0; i < 16; ++i) F[i >> 2][i & 3] = fa * i;
#define MF(cc, aa, bb) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(cc) : "v"(aa), "v"(bb));
#define VF(j) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(x[(j) & 7]) : "v"(fb));
  for (int it = 0; it < N_ITER; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        MF(c[i], fa, fb)
#pragma unroll
        for (int j = 0; j < NV; ++j) VF(i + j)
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        MF(c[0], fa, fb)
#pragma unroll
        for (int j = 0; j < NV; ++j) VF(i + j)
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // (the compiler inserts the wait states the hazard needs)
        c[(i + 1) & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, c[i][0], c[(i + 1) & 7], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) VF(i + j)
      }
    } else {
      f4 Y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) Y[q] = c[q];
      dct_fwd(F, E, Y);
#pragma unroll
      for (int q = 0; q < 4; ++q) { c[q] = Y[q]; F[q][0] = Y[q][1]; }
#pragma unroll
      for (int j = 0; j < NV; ++j) VF(j)
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE, int NV> void run(float* d, int wps) {
  const int blocks = 1024 * wps;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_rate<MODE, NV><<<blocks, 64>>>(d, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(a); k_rate<MODE, NV><<<blocks, 64>>>(d, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double per = MODE == 3 ? 32.0 : 8.0;
  const double n = (double)wps * N_ITER * per;
  static const char* nm[4] = {"independent", "dependent accumulator", "result -> next B operand", "forward transform (32)"};
  printf("%-26s waves/SIMD %d fillers %2d%s: %.3f ms -> %.1f cycles per MFMA per SIMD at 2.4 GHz\n", nm[MODE], wps, NV,
         MODE == 3 ? " per transform" : " per MFMA", ms, ms * 1e-3 / n * 2.4e9);
}

static double cdct(int u, int x) { return (u == 0 ? sqrt(1.0 / 8) : sqrt(2.0 / 8)) * cos(M_PI * (2 * x + 1) * u / 16.0); }

int main() {
  // ---- 1
  float ha[64], hb[64], hc[256], hd[256], *da, *db, *dc, *dd;
  srand(1);
  for (int i = 0; i < 64; ++i) { ha[i] = 1.f + rand() / (float)RAND_MAX; hb[i] = 1.f + rand() / (float)RAND_MAX; }
  for (int i = 0; i < 256; ++i) hc[i] = rand() / (float)RAND_MAX;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(da, db, dc, dd);
  hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0, fused = 0, unfused = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const float A = ha[4 * (l >> 2) + v], B = hb[l], C = hc[l * 4 + v];
      const float f = fmaf(A, B, C);
      volatile float pr = A * B;
      const float u = pr + C;
      const float got = hd[l * 4 + v];
      if (got == f) ++fused;
      if (got == u) ++unfused;
      if (got != f && got != u) ++bad;
    }
  printf("layout D[reg v][lane l] = A[lane 4*(l>>2)+v] * B[lane l] + C[reg v][lane l]: %s (%d of 256 off); equals fmaf %d, equals mul-then-add %d of 256\n",
         bad ? "WRONG" : "confirmed", bad, fused, unfused);
  // ---- 2
  float hp[1024], hbasis[64], hcoef[1024], hback[1024], *dp, *dbs, *dco, *dbk;
  for (int i = 0; i < 1024; ++i) hp[i] = 255.f * rand() / (float)RAND_MAX;
  for (int u = 0; u < 8; ++u) for (int x = 0; x < 8; ++x) hbasis[u * 8 + x] = (float)cdct(u, x);
  hipMalloc(&dp, 4096); hipMalloc(&dbs, 256); hipMalloc(&dco, 4096); hipMalloc(&dbk, 4096);
  hipMemcpy(dp, hp, 4096, hipMemcpyHostToDevice); hipMemcpy(dbs, hbasis, 256, hipMemcpyHostToDevice);
  k_dct<<<1, 64>>>(dp, dbs, dco, dbk);
  hipMemcpy(hcoef, dco, 4096, hipMemcpyDeviceToHost); hipMemcpy(hback, dbk, 4096, hipMemcpyDeviceToHost);
  double emax = 0, rmax = 0;
  for (int p = 0; p < 16; ++p)
    for (int u = 0; u < 8; ++u)
      for (int v = 0; v < 8; ++v) {
        double s = 0;
        for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) s += cdct(u, y) * cdct(v, x) * hp[p * 64 + y * 8 + x];
        emax = fmax(emax, fabs(s - hcoef[p * 64 + u * 8 + v]));
        rmax = fmax(rmax, fabs(hback[p * 64 + u * 8 + v] - hp[p * 64 + u * 8 + v]));
      }
  printf("separable folded DCT on 4x4x1 blocks: max |coef - double DCT| = %.3g, max |inverse(forward) - patch| = %.3g (0..255 data)\n", emax, rmax);
  // ---- 3
  float* d; hipMalloc(&d, 4 * 64 * 8192);
  for (int w : {1, 2, 3, 4}) {
    run<0, 0>(d, w); run<0, 1>(d, w); run<0, 2>(d, w); run<0, 4>(d, w);
    run<1, 0>(d, w); run<1, 1>(d, w); run<1, 2>(d, w);
    run<2, 0>(d, w); run<2, 2>(d, w);
    run<3, 0>(d, w); run<3, 16>(d, w); run<3, 32>(d, w); run<3, 64>(d, w);
  }
  return 0;
}
