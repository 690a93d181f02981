// Micro-benchmark: LDS read rates of a sliding window (lane l reads floats l + u, u = 0..) on gfx950:
// 4 x ds_read_b32 against one unaligned ds_read_b128 / two ds_read_b64 of the same 4 floats.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rates lds_rates.hip && ./lds_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 500
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int shift) {
  __shared__ float t[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) t[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned base = (unsigned)(size_t)t + 4 * (lane + shift);  // LDS byte address: dword-aligned only
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int i = 0; i < N_ITER; ++i) {
    const unsigned a = base + 64 * (i & 15);
    if (MODE == 0) {
      float x0, x1, x2, x3, y0, y1, y2, y3;
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                   "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(a));
      s0 += x0 + y0; s1 += x1 + y1; s2 += x2 + y2; s3 += x3 + y3;
    } else if (MODE == 1) {
      f4 x, y;
      asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(x), "=v"(y) : "v"(a));
      s0 += x.x + y.x; s1 += x.y + y.y; s2 += x.z + y.z; s3 += x.w + y.w;
    } else if (MODE == 2) {
      f2 x, y, z, w;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                   : "=v"(x), "=v"(y), "=v"(z), "=v"(w) : "v"(a));
      s0 += x.x + z.x; s1 += x.y + z.y; s2 += y.x + w.x; s3 += y.y + w.y;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3;
}
template <int MODE> void run(const char* name, float* d, int blocks, int shift) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, shift); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, shift); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipError_t e = hipGetLastError();
  double floats = (double)blocks * 4 * N_ITER * 8;  // 8 floats per lane per iteration
  printf("%-30s shift=%d blocks=%5d  %.3f ms  %.2f cycles per 8 floats/lane per CU at 2.4 GHz  (%s)\n", name, shift, blocks, ms,
         256.0 * 2.4e9 / (floats / 8 / (ms * 1e-3)), hipGetErrorString(e));
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 8192);
  for (int shift : {0, 1, 3}) {
    run<0>("8 x ds_read_b32", d, 2048, shift); run<1>("2 x ds_read_b128 (unaligned)", d, 2048, shift); run<2>("4 x ds_read_b64 (unaligned)", d, 2048, shift);
  }
  return 0;
}
