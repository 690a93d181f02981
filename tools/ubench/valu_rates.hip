// Micro-benchmark: issue rate of the VALU forms the group kernel is made of
// (plain FMA, DPP FMA, v_mov_dpp, packed FMA, LDS read/write) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 2000
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float c = a;
  for (int i = 0; i < N_ITER; ++i) {
    if (MODE == 0) {
      asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
                   "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    } else if (MODE == 1) {
#define D " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      asm volatile("v_fmac_f32_dpp %0, %1, %8" D "v_fmac_f32_dpp %1, %2, %8" D "v_fmac_f32_dpp %2, %3, %8" D "v_fmac_f32_dpp %3, %4, %8" D
                   "v_fmac_f32_dpp %4, %5, %8" D "v_fmac_f32_dpp %5, %6, %8" D "v_fmac_f32_dpp %6, %7, %8" D "v_fmac_f32_dpp %7, %0, %8" D
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    } else if (MODE == 2) {
      asm volatile("v_pk_fma_f32 %0, %2, %4, %0\n v_pk_fma_f32 %1, %3, %4, %1\n v_pk_fma_f32 %2, %0, %4, %2\n v_pk_fma_f32 %3, %1, %4, %3\n"
                   "v_pk_fma_f32 %0, %2, %4, %0\n v_pk_fma_f32 %1, %3, %4, %1\n v_pk_fma_f32 %2, %0, %4, %2\n v_pk_fma_f32 %3, %1, %4, %3"
                   : "+v"(*(double*)&x0), "+v"(*(double*)&x2), "+v"(*(double*)&x4), "+v"(*(double*)&x6) : "v"(*(double*)&c));
    } else if (MODE == 3) {
#define H " row_half_mirror row_mask:0xf bank_mask:0xf\n"
      asm volatile("v_mov_b32_dpp %0, %1" H "v_mov_b32_dpp %1, %2" H "v_mov_b32_dpp %2, %3" H "v_mov_b32_dpp %3, %4" H
                   "v_mov_b32_dpp %4, %5" H "v_mov_b32_dpp %5, %6" H "v_mov_b32_dpp %6, %7" H "v_mov_b32_dpp %7, %0" H
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 4) {
      asm volatile("v_fma_f32 %0, %8, %0, %1\n v_fma_f32 %1, %8, %1, %2\n v_fma_f32 %2, %8, %2, %3\n v_fma_f32 %3, %8, %3, %4\n"
                   "v_fma_f32 %4, %8, %4, %5\n v_fma_f32 %5, %8, %5, %6\n v_fma_f32 %6, %8, %6, %7\n v_fma_f32 %7, %8, %7, %0"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(const char* name, float* d, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr = (double)blocks * 4 /*waves*/ * N_ITER * 8;
  printf("%-28s blocks=%5d  %.3f ms  %.1f G wave-instr/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, blocks, ms,
         instr / ms / 1e6, 1024.0 * 2.4e9 / (instr / (ms * 1e-3)));
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 8192);
  for (int blocks : {256, 1024, 2048}) {
    run<0>("v_fmac_f32", d, blocks); run<4>("v_fma_f32 (VOP3)", d, blocks); run<1>("v_fmac_f32_dpp quad_perm", d, blocks);
    run<3>("v_mov_b32_dpp half_mirror", d, blocks); run<2>("v_pk_fma_f32", d, blocks);
  }
  return 0;
}
