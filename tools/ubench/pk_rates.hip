// Micro-benchmark: issue rate of packed f32 VALU forms on gfx950 (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) against
// plain v_add_f32, with 8 independent accumulator chains per wavefront so that latency is not what is measured.
//   hipcc --offload-arch=gfx950 -O3 -o pk_rates pk_rates.hip && ./pk_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 2000
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a) {
  f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  f2 c = {a, a * 1.5f}, d = {a * 0.5f, a * 0.25f};
  for (int i = 0; i < N_ITER; ++i) {
    if (MODE == 0) {  // 16 plain adds
      asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                   "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7"
                   : "+v"(x0.x), "+v"(x1.x), "+v"(x2.x), "+v"(x3.x), "+v"(x4.x), "+v"(x5.x), "+v"(x6.x), "+v"(x7.x) : "v"(c.x));
      asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                   "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7"
                   : "+v"(x0.y), "+v"(x1.y), "+v"(x2.y), "+v"(x3.y), "+v"(x4.y), "+v"(x5.y), "+v"(x6.y), "+v"(x7.y) : "v"(c.y));
    } else if (MODE == 1) {  // 8 packed adds = the same 16 additions
      asm volatile("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n"
                   "v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    } else if (MODE == 2) {
      asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n"
                   "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    } else if (MODE == 3) {
      asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                   "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c), "v"(d));
    } else if (MODE == 4) {  // packed add whose second source is another accumulator (two 64-bit VGPR sources that change)
      asm volatile("v_pk_add_f32 %0, %1, %0\n v_pk_add_f32 %2, %3, %2\n v_pk_add_f32 %4, %5, %4\n v_pk_add_f32 %6, %7, %6\n"
                   "v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %3, %8, %3\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %7, %8, %7"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    } else if (MODE == 5) {  // packed sub + packed mul + packed add: the matcher's (c - t)^2 accumulate, two pixels at a time
      asm volatile("v_pk_add_f32 %4, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %4, %4, %4\n v_pk_add_f32 %1, %4, %1\n"
                   "v_pk_add_f32 %5, %2, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %5, %5, %5\n v_pk_add_f32 %3, %5, %3\n"
                   "v_pk_add_f32 %6, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %6, %6, %6"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
    }
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE> void run(const char* name, float* d, int blocks, int ninstr) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr = (double)blocks * 4 /*waves*/ * N_ITER * ninstr;
  printf("%-34s blocks=%5d  %.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz\n", name, blocks, ms, 1024.0 * 2.4e9 / (instr / (ms * 1e-3)));
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 8192);
  for (int blocks : {1024, 2048}) {
    run<0>("v_add_f32 x16", d, blocks, 16); run<1>("v_pk_add_f32 x8 (same additions)", d, blocks, 8);
    run<2>("v_pk_mul_f32 x8", d, blocks, 8); run<3>("v_pk_fma_f32 x8", d, blocks, 8);
    run<4>("v_pk_add_f32 x8, two VGPR pairs", d, blocks, 8); run<5>("pk sub/mul/add mix x8", d, blocks, 8);
  }
  return 0;
}
