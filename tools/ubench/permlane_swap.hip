// What v_permlane32_swap / v_permlane16_swap do on gfx950 (the reduction of k_group8m.h relies on it):
//   hipcc --offload-arch=gfx950 -O3 -o permlane_swap permlane_swap.hip && ./permlane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  const unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 [0]", "swap32 [1]", "swap16 [0]", "swap16 [1]"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int l = 0; l < 64; l += 8) printf(" %u", h[64 * v + l]);
    printf("\n");
  }
  return 0;
}
