// k_frame.h — whole-frame, HBM-bound kernels: layout change, validity map,
// normalisation, colour transform, bicubic warp.
#pragma once
#include "nlk_common.h"

// The layout work of a frame call in one pass over the pixels: up to three HWC images -> planar, the
// row test of the validity map on the previous frame's channel 0 (see k_nan_cols), and the
// accumulator cleared (whole-frame calls; a strip's accumulator belongs to the caller).
// CH: the channel count at compile time (3: a pixel is ONE 12-byte load per image instead of three 4-byte ones), 0 = `ch`
template <int CH>
__global__ void __launch_bounds__(256)
k_layout(const float* __restrict__ cur, float* __restrict__ pl_cur, const float* __restrict__ prev,
         float* __restrict__ pl_prev, const float* __restrict__ basic, float* __restrict__ pl_basic,
         uint8_t* __restrict__ rowok, float* __restrict__ acc_zero, int w, int h, int ch, int psz, int planar,
         int y0,    // (rows y0 .. y0 + gridDim.y - 1: a frame that arrives from the host in row bands)
         uint32_t* __restrict__ zero_word,    // one more word to clear (the wide-window queue's length), or nullptr
         float* __restrict__ pl_diff) {       // smoother calls: planar prev - cur (k_group8m's pass B reads one image instead of two), or nullptr
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = y0 + blockIdx.y;
  if (zero_word && x == 0 && blockIdx.y == 0) *zero_word = 0u;
  if (x >= w) return;
  const size_t npix = (size_t)w * h, i = (size_t)y * w + x;
  float own = 0.f;   // channel 0 of the previous frame's pixel (the row test below)
  if constexpr (CH == 3) {
    struct __attribute__((packed, aligned(4))) P3 { float v[3]; };
    const P3 a = reinterpret_cast<const P3*>(cur)[i];
    P3 b = a, d = a;
    if (prev) b = reinterpret_cast<const P3*>(prev)[i];
    if (basic) d = reinterpret_cast<const P3*>(basic)[i];
    own = b.v[0];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pl_cur[c * npix + i] = a.v[c];
      if (prev) pl_prev[c * npix + i] = b.v[c];
      if (basic) pl_basic[c * npix + i] = d.v[c];
      if (pl_diff) pl_diff[c * npix + i] = b.v[c] - a.v[c];
    }
  } else {
    if (prev) own = prev[i * ch];
    if (planar) {  // (one channel: planar == interleaved, the images are used in place)
      for (int c = 0; c < ch; ++c) {
        pl_cur[c * npix + i] = cur[i * ch + c];
        if (prev) pl_prev[c * npix + i] = prev[i * ch + c];
        if (basic) pl_basic[c * npix + i] = basic[i * ch + c];
        if (pl_diff) pl_diff[c * npix + i] = prev[i * ch + c] - cur[i * ch + c];
      }
    }
  }
  if (prev) {
#ifdef NLK_LAYOUT_ROWOK_LOADS   // (before round 6: psz loads per pixel, all but one of them for values its neighbours hold)
    uint8_t ok = (x + psz <= w);
    if (ok)
      for (int j = 0; j < psz; ++j) {
        const float v = prev[(i + j) * ch];
        if (v != v) ok = 0;
      }
    rowok[i] = ok;
#else
    // row test of the validity map: no NaN in channel 0 of the psz pixels from x on. A wavefront covers 64 consecutive
    // pixels of a row (blockDim.x = 256: four of them), so the test is a window of psz bits in the ballot of the
    // wavefront's own NaN flags, continued by the flags of the psz - 1 pixels behind its last one (psz <= 32).
    const int lane = threadIdx.x & 63;
    const int xh = x + 64;   // the halo pixel this lane looks at: the wavefront's first pixel + 64 + lane
    const bool halo_nan = lane < psz - 1 && xh < w && prev[((size_t)y * w + xh) * ch] != prev[((size_t)y * w + xh) * ch];
    const uint64_t m_lo = __ballot(own != own), m_hi = __ballot(halo_nan);
    const uint64_t win = (m_lo >> lane) | (lane ? m_hi << (64 - lane) : 0ull);
    rowok[i] = (x + psz <= w) && (win & ((1ull << psz) - 1ull)) == 0ull;
#endif
  }
  if (acc_zero)
    for (int c = 0; c <= ch; ++c) acc_zero[c * npix + i] = 0.f;
}

// valid[y][x] = 1 iff the psz x psz patch of plane 0 of the previous frame with
// origin (x,y) lies in the image and holds no NaN
// (reference: src/nlkalman.c:605-609, 725-730 — only channel 0 is tested).
// Two separable passes over a byte map: rows (in k_layout) then columns.
__global__ void k_nan_cols(const uint8_t* __restrict__ rowok, uint8_t* __restrict__ valid,
                           int w, int h, int psz, int y0) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = y0 + blockIdx.y;
  if (x >= w) return;
  uint8_t ok = (y + psz <= h);
  if (ok)
    for (int i = 0; i < psz; ++i) ok &= rowok[(size_t)(y + i) * w + x];
  valid[(size_t)y * w + x] = ok;
}
// the same on four pixels per thread (w % 4 == 0: rows of 32-bit words of 0 / 1 bytes)
__global__ void k_nan_cols4(const uint32_t* __restrict__ rowok, uint32_t* __restrict__ valid,
                            int w4, int h, int psz, int y0) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = y0 + blockIdx.y;
  if (x >= w4) return;
  uint32_t ok = (y + psz <= h) ? 0x01010101u : 0u;
  if (ok)
    for (int i = 0; i < psz; ++i) ok &= rowok[(size_t)(y + i) * w4 + x];
  valid[(size_t)y * w4 + x] = ok;
}

// out = acc_c / acc_w where acc_w > 1e-6 else the input frame
// (reference: src/nlkalman.c:939-942, 1853-1856; the double literal 1e-6 there
// compares like 1e-6f against a float weight)
// CH: the channel count at compile time (3: a pixel leaves as ONE 12-byte store), 0 = `ch`
template <int CH>
__global__ void k_normalize(float* __restrict__ out, const float* __restrict__ acc,
                            const float* __restrict__ cur_hwc, int w, int h, int ch,
                            int y0, int y1) {
  const size_t npix = (size_t)w * h;
  for (size_t i = (size_t)y0 * w + blockIdx.x * blockDim.x + threadIdx.x;
       i < (size_t)y1 * w; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (CH == 3) {
      struct __attribute__((packed, aligned(4))) P3 { float v[3]; };
      const float a = acc[(size_t)3 * npix + i];
      P3 o;
      if (a > 1e-6f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o.v[c] = acc[(size_t)c * npix + i] / a;
      } else {
        o = reinterpret_cast<const P3*>(cur_hwc)[i];  // (the input frame is read only where it is needed)
      }
      reinterpret_cast<P3*>(out)[i] = o;
    } else {
      const float a = acc[(size_t)ch * npix + i];
      for (int c = 0; c < ch; ++c)  // (the input frame is read only where it is needed)
        out[i * ch + c] = a > 1e-6f ? acc[(size_t)c * npix + i] / a : cur_hwc[i * ch + c];
    }
  }
}

// dst += src (accumulator halo rows received from a neighbouring strip)
__global__ void k_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}

// reference: src/nlkalman.c:92-110 (in place, ch == 3)
__global__ void k_rgb2opp(float* __restrict__ im, size_t npix) {
  const float a = 1.f / sqrtf(3.f), b = 1.f / sqrtf(2.f);
  const float c = 2.f * a * sqrtf(2.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix;
       i += (size_t)gridDim.x * blockDim.x) {
    float* p = im + 3 * i;
    const float r = p[0], g = p[1], bl = p[2];
    p[0] = a * (r + g + bl);
    p[1] = b * (r - bl);
    p[2] = c * (0.25f * r - 0.5f * g + 0.25f * bl);
  }
}

// reference: src/nlkalman.c:112-130
__global__ void k_opp2rgb(float* __restrict__ im, size_t npix) {
  const float a = 1.f / sqrtf(3.f), b = 1.f / sqrtf(2.f);
  const float c = a / b;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix;
       i += (size_t)gridDim.x * blockDim.x) {
    float* p = im + 3 * i;
    const float y = p[0], u = p[1], v = p[2];
    p[0] = a * y + b * u + 0.5f * c * v;
    p[1] = a * y - c * v;
    p[2] = a * y - b * u + 0.5f * c * v;
  }
}

// Keys cubic with the reference's mixed float/double evaluation
// (reference: src/nlkalman.c:36-41)
__device__ inline float nlk_cubic(const float v[4], float x) {
  return (float)(v[1] + 0.5 * x * (v[2] - v[0] +
                 x * (2.0 * v[0] - 5.0 * v[1] + 4.0 * v[2] - v[3] +
                      x * (3.0 * (v[1] - v[2]) + v[3] - v[0]))));
}

// reference: src/nlkalman.c:29-33, 43-88 — one thread per output pixel. The 16 taps are gathered first, all
// channels of a tap together (CH = 3: one 12-byte load per tap instead of three 4-byte ones; CH = 0: any channel
// count, tap by tap), then every channel is interpolated in the reference's order: columns with fy, then the row with fx.
template <int CH>
__global__ void k_warp_bicubic(float* __restrict__ imw, const float* __restrict__ im,
                               const float* __restrict__ of, const float* __restrict__ msk,
                               int w, int h, int ch) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const size_t pix = (size_t)x + (size_t)y * w;
  float* o = imw + pix * ch;
  if (msk && msk[pix] != 0.f) {
    for (int c = 0; c < ch; ++c) o[c] = __builtin_nanf("");
    return;
  }
  float xw = x + of[pix * 2 + 0];
  float yw = y + of[pix * 2 + 1];
  xw -= 1;
  yw -= 1;
  const int ix = (int)floorf(xw), iy = (int)floorf(yw);
  const float fx = xw - ix, fy = yw - iy;
  if (CH > 0) {
    float t[4][4][CH > 0 ? CH : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sx = ix + i, sy = iy + j;
        const bool in = !(sx < 0 || sx >= w || sy < 0 || sy >= h);
        const float* p = im + ((size_t)(in ? sx : 0) + (size_t)(in ? sy : 0) * w) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float v = p[c];
          t[i][j][c] = in ? v : __builtin_nanf("");
        }
      }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float col[4] = {t[i][0][c], t[i][1][c], t[i][2][c], t[i][3][c]};
        v[i] = nlk_cubic(col, fy);
      }
      o[c] = nlk_cubic(v, fx);
    }
    return;
  }
  for (int c = 0; c < ch; ++c) {
    float v[4];
    for (int i = 0; i < 4; ++i) {
      float col[4];
      for (int j = 0; j < 4; ++j) {
        const int sx = ix + i, sy = iy + j;
        col[j] = (sx < 0 || sx >= w || sy < 0 || sy >= h)
                     ? __builtin_nanf("")
                     : im[((size_t)sx + (size_t)sy * w) * ch + c];
      }
      v[i] = nlk_cubic(col, fy);
    }
    o[c] = nlk_cubic(v, fx);
  }
}
