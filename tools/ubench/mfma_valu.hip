// Micro-benchmark: does f32 VALU work issue in the shadow of v_mfma_f32_16x16x4_f32 on
// gfx950? One wave per SIMD (or W waves) runs a loop of 4 independent MFMAs, each followed
// by NV independent v_fma_f32; the time per MFMA is printed against NV.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define N_ITER 1000
template <int NV>
__global__ void __launch_bounds__(64) k(float* out, float a) {
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  const float fa = threadIdx.x * 0.001f, fb = a;
  for (int it = 0; it < N_ITER; ++it) {
#define MF(c) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(fa), "v"(fb));
#define VF(j) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(x[(j) & 7]) : "v"(fb));
    MF(c0)
#pragma unroll
    for (int j = 0; j < NV; ++j) VF(j)
    MF(c1)
#pragma unroll
    for (int j = 0; j < NV; ++j) VF(j + 3)
    MF(c2)
#pragma unroll
    for (int j = 0; j < NV; ++j) VF(j + 5)
    MF(c3)
#pragma unroll
    for (int j = 0; j < NV; ++j) VF(j + 7)
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s + c0[0] + c1[1] + c2[2] + c3[3];
}
template <int NV> void run(float* d, int waves_per_simd) {
  const int blocks = 1024 * waves_per_simd;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<NV><<<blocks, 64>>>(d, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(a); k<NV><<<blocks, 64>>>(d, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double mfma_per_simd = (double)waves_per_simd * N_ITER * 4;
  printf("waves/SIMD %d  NV %2d: %.3f ms  -> %.1f ns per MFMA slot per SIMD (%.1f cycles at 2.4 GHz)\n",
         waves_per_simd, NV, ms, ms * 1e6 / mfma_per_simd, ms * 1e-3 / mfma_per_simd * 2.4e9);
}
int main() {
  float* d; hipMalloc(&d, 4 * 64 * 8192);
  for (int w : {1, 2, 4}) {
    run<0>(d, w); run<2>(d, w); run<4>(d, w); run<6>(d, w); run<8>(d, w); run<12>(d, w); run<16>(d, w);
  }
  return 0;
}
