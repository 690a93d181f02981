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
};

static __host__ __device__ inline uint32_t nlk_pack_xy(int x, int y) {
  return (uint32_t)x | ((uint32_t)y << 16);
}
static __host__ __device__ inline int nlk_x(uint32_t p) { return (int)(p & 0xffffu); }
static __host__ __device__ inline int nlk_y(uint32_t p) { return (int)(p >> 16); }
