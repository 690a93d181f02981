// Micro-benchmark: adding a wavefront's 16 x (2 rows x 8 pixels) patch pieces into an LDS accumulator tile on gfx950 -
// ds_add_f32 (LDS float atomic, no return) against read / fma / write, with and without lanes that meet on one
// address inside an instruction.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip && ./lds_atomic
// Lane = 4 * patch + i adds pixels (row i, columns 0..7) and (row 7 - i, columns 0..7) of its patch; patch p sits at
// tile offset off[p]. MODE 0: read-modify-write (only valid when the 16 patches of an instruction do not overlap);
// MODE 1: v_mul + ds_add_f32; OVERLAP 0: patches in 4 planes x 4 far-apart positions; 1: the 4 patches of a plane
// one pixel apart (every instruction has lanes meeting on an address); 2: all 16 patches at ONE position.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 400
#define RWP 28
#define PLANE (RWP * 24)
template <int MODE, int OVERLAP>
__global__ void __launch_bounds__(64) k(float* out, float w, int sh) {
  __shared__ float tile[4 * PLANE];
  for (int i = threadIdx.x; i < 4 * PLANE; i += 64) tile[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x, p = lane >> 2, i = lane & 3;
  const int pl = p & 3, mem = p >> 2;
  int off;
  if (OVERLAP == 0) off = pl * PLANE + (mem >> 1) * 9 * RWP + (mem & 1) * 12;
  else if (OVERLAP == 1) off = pl * PLANE + mem * RWP + mem;
  else off = 5 * RWP + 3;
  off += sh;  // (run-time: the position of a patch in the tile is only dword-aligned)
  float* ra = tile + off + i * RWP;
  float* rb = tile + off + (7 - i) * RWP;
  float px[16];
  for (int c = 0; c < 16; ++c) px[c] = lane + c;
  for (int it = 0; it < N_ITER; ++it) {
    if (MODE == 0) {
      float old[16];
#pragma unroll
      for (int c = 0; c < 8; ++c) { old[c] = ra[c]; old[8 + c] = rb[c]; }
#pragma unroll
      for (int c = 0; c < 8; ++c) { ra[c] = fmaf(w, px[c], old[c]); rb[c] = fmaf(w, px[8 + c], old[8 + c]); }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(ra + c), w * px[c], 0, 0, false);
        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(rb + c), w * px[8 + c], 0, 0, false);
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(px[c]));
  }
  __syncthreads();
  float s = 0;
  for (int j = threadIdx.x; j < 4 * PLANE; j += 64) s += tile[j];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE, int OVERLAP> void run(float* d, int wps) {
  const int blocks = 1024 * wps;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, OVERLAP><<<blocks, 64>>>(d, 0.5f, 1); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE, OVERLAP><<<blocks, 64>>>(d, 0.5f, 1); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // 1024 * wps workgroups over 256 CUs = 4 * wps per CU, one after the other per occupancy slot
  const double batches_per_cu = 4.0 * wps * N_ITER;
  printf("%-18s overlap %d, %d waves/SIMD: %.3f ms -> %.0f cycles per 16-patch batch per CU (%.1f per wave-instruction pair)\n",
         MODE ? "v_mul + ds_add_f32" : "read / fma / write", OVERLAP, wps, ms, ms * 1e-3 * 2.4e9 / batches_per_cu,
         ms * 1e-3 * 2.4e9 / batches_per_cu / 16);
}
int main() {
  float* d; hipMalloc(&d, 4 * 64 * 8192);
  for (int w : {1, 2, 3}) {
    run<0, 0>(d, w); run<1, 0>(d, w); run<1, 1>(d, w); run<1, 2>(d, w);
  }
  return 0;
}
