// nlk_common.h — shared definitions of the gfx950 kernels (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Geometry + parameters of one frame (or row strip) call, passed by value to
// every kernel. Images are planar on the device: plane c at img + c*h*w.
struct NlkGeom {
  int w, h, ch;            // strip size (h includes the search halo rows)
  int psz, step, p2, E;    // patch size, grid step psz/2, psz^2, ch*psz^2
  int ngx, ngy, oy;        // target grid: x = gx*step, y = oy + gy*step
  int wsz_x, wsz_t;        // search radii (reference: src/nlkalman.c:637)
  int npx, npt, ntagg;     // k for spatial / temporal targets, group size
  int R;                   // reach of a group on the patch grid: max(wsz)/step
  int kmax;                // row stride of the top-k record
  int gstride;             // row stride of the group-coordinate record (>= 1)
  int have_prev, have_basic, smoother;
  float sigma2, beta_x, beta_t;
};

// per-target record written by the matching kernel
struct NlkTarget {
  int nsel;   // number of kept candidates (0: nothing to do)
  int np0;    // kept candidates with a valid previous patch
  int nagg;   // group members that are filtered and aggregated
  int flags;  // bit0: prev_p, bit1: group marks the processed-mask
  uint64_t vbits[2];  // bit i: kept candidate i (< 128) has a valid previous patch
};

static __host__ __device__ inline uint32_t nlk_pack_xy(int x, int y) {
  return (uint32_t)x | ((uint32_t)y << 16);
}
static __host__ __device__ inline int nlk_x(uint32_t p) { return (int)(p & 0xffffu); }
static __host__ __device__ inline int nlk_y(uint32_t p) { return (int)(p >> 16); }

// XCD-aware tile order. Workgroups are dealt round-robin over the 8 XCDs (blocks
// b and b+8 share an XCD and its L2, MI355X_MICROARCH.md), so block b takes tile
// (b % 8) * ceil(n/8) + b / 8: every XCD works through one contiguous band of
// tiles and re-reads of neighbouring rows hit its own L2. The grid is launched
// with 8 * ceil(n/8) blocks; the mapping is a bijection onto [0, 8*ceil(n/8)),
// indices >= n have no tile. Only speed depends on the actual placement.
static __device__ inline int nlk_xcd_tile(int b, int n) {
  const int per = (n + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}
static inline int nlk_xcd_grid(int n) { return ((n + 7) >> 3) << 3; }

// LDS written by some lanes of a wavefront and read by others of the SAME wavefront: order the
// accesses without a workgroup barrier
__device__ inline void nlk_wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
}
// The same without the wait: the LDS executes the DS instructions of one wavefront in the order
// they were issued, so a read issued after a write of the same wavefront returns the written
// data; only the COMPILER must be kept from moving the read above the write (the wait for the
// data itself is the ordinary lgkmcnt wait in front of its first use).
__device__ inline void nlk_wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
