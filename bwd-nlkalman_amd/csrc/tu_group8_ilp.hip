// tu_group8_ilp.hip - the instantiations of k_group8m that respond to the compiler's instruction-level-parallelism
// scheduler (Makefile: -mllvm -amdgpu-sched-strategy=max-ilp for THIS unit only). Measured at 1080p RGB, group ms,
// default scheduler -> max-ilp (profiles/README.md rounds 5, 6): the filter in the hybrid form (first frames since
// round 6) 0.743 -> 0.723, first frame 0.974 -> 0.957; the filter in the separable form (temporal frames since round 6)
// 0.667 -> 0.662, FLT2 0.580 -> 0.576; the others lose under it or do not respond - FLT2 in the Kronecker form
// 0.611 -> 0.633, one channel FLT1 0.394 -> 0.405, SMO1 0.549 -> 0.563, RGB SMO1 in the hybrid form 1.101 / 1.098 - and
// stay in tu_group8.hip.
#include "k_group8m.h"
#include "nlk_internal.h"

// the kernel for (channels, smoother, DCT form) if this unit holds it, else nullptr
const void* nlk_group8m_ilp_kernel(int ch, bool smoother, int sep) {
  if (ch == 3 && !smoother && sep == 2) return (const void*)k_group8m<3, false, 2, 1>;
  if (ch == 3 && !smoother && sep == 6) return (const void*)k_group8m<3, false, 6, 1>;
  return nullptr;
}
