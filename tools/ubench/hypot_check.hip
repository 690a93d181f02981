// Exhaustive-style check of csrc/tv_hypot.h on the GPU: for 2^34 pseudo-random float pairs
// (flow-gradient magnitudes, tiny, huge, equal, zero) it compares nlk_tv_hypot with
// (float)sqrt((double)x*x + (double)y*y), counts mismatches, counts how often the exact path
// is taken and records the largest relative error of the fast root against the exact one.
//   hipcc --offload-arch=gfx950 -O3 -I../../bwd-nlkalman_amd/csrc -o hypot_check hypot_check.hip && ./hypot_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include "tv_hypot.h"

__device__ __forceinline__ unsigned mix(unsigned long long& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  unsigned long long z = s;
  z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33;
  return (unsigned)(z >> 16);
}

__device__ __forceinline__ float draw(unsigned long long& st, int kind) {
  const unsigned a = mix(st), b = mix(st);
  const float m = 1.f + (a & 0x7fffff) * 1.1920929e-7f;  // [1, 2)
  int e;
  switch (kind) {
    case 0: e = (int)(b % 24) - 16; break;      // 2^-16 .. 2^7: flow gradients
    case 1: e = (int)(b % 250) - 125; break;    // the whole normal range
    case 2: e = -126 - (int)(b % 23); break;    // denormal floats
    default: e = (int)(b % 8) - 4; break;
  }
  const float v = ldexpf(m, e);
  return (b & 0x80000000u) ? -v : v;
}

__global__ void k_check(unsigned long long seed, int per_thread, unsigned long long* bad, unsigned long long* slow,
                        unsigned long long* errbits) {
  unsigned long long st = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull;
  unsigned long long nbad = 0, nslow = 0;
  double emax = 0.0;
  for (int i = 0; i < per_thread; ++i) {
    const int kind = i & 7;
    float x = draw(st, kind < 5 ? 0 : kind - 4), y = draw(st, kind < 4 ? 0 : kind == 4 ? 3 : kind - 4);
    if ((i & 63) == 9) y = 0.f;
    if ((i & 63) == 17) { x = 0.f; if (i & 64) y = 0.f; }
    if ((i & 63) == 33) y = x;
    const float fast = nlk_tv_hypot(x, y);
    const double s = (double)x * x + (double)y * y;
    const double ex = sqrt(s);
    const float ref = (float)ex;
    if (__float_as_uint(fast) != __float_as_uint(ref)) ++nbad;
    if (s != 0.0) {
      const double g = nlk_tv_sqrt_fast(s);
      const double e = fabs(g - ex) / ex;
      emax = e > emax ? e : emax;
      const unsigned lo = (unsigned)__double_as_longlong(g) & 0x1fffffffu;
      if (lo - (0x10000000u - NLK_TV_HYPOT_BAND) < 2u * NLK_TV_HYPOT_BAND) ++nslow;
    }
  }
  atomicAdd(bad, nbad);
  atomicAdd(slow, nslow);
  atomicMax(errbits, (unsigned long long)__double_as_longlong(emax));
}

int main() {
  unsigned long long *d, h[3] = {0, 0, 0};
  hipMalloc(&d, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int blocks = 4096, threads = 256, per_thread = 1 << 14, rounds = 1;
  for (int r = 0; r < rounds; ++r)
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull + r * 977ull, per_thread, d, d + 1, d + 2);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double emax;
  memcpy(&emax, &h[2], 8);
  const double n = (double)blocks * threads * per_thread * rounds;
  printf("%.3g pairs: %llu mismatches, exact path taken for %llu (%.2e), largest relative error of the fast root 2^%.2f\n",
         n, h[0], h[1], h[1] / n, emax > 0 ? log2(emax) : -999.0);
  return h[0] != 0;
}
