// k_group12.h — group processing for 12x12 patches (BASELINE config C3), the
// register/DPP fast path (reference: src/nlkalman.c:713-932 filter, :1603-1845
// smoother). Same idea as k_group8.h with the layout adapted to 12 rows:
//
//   lane l = 16*c + u : row u (< 12) of channel c, one DPP row of 16 lanes per
//   channel; the image patch row lives in registers a[12], the previous-frame
//   patch row in b[12] of the SAME lane. Hence
//   * row pass of the DCT: register arithmetic (even/odd split, 12 + 72 ops),
//   * column pass: a transposition of the channel's 12x12 tile through LDS and the
//     register transform again (nlk_transpose12 below). The first version
//     (NLK_GROUP12_DPP) fed 15 `row_ror:k` rotations of the DPP row to
//     v_fmac_f32_dpp, with the per-lane coefficient of rotation k looked up once
//     per kernel by rotating the lane index itself: half of the kernel's time,
//   * Welford statistics of both patches, the transition variance (b - a)^2,
//     gains and shrinkage are all lane-local — no partner exchange at all,
//   * pass B transforms two group members per step (one in a[], one in b[]).
// Aggregation: private LDS tile per wavefront, plain read-modify-write, flushed
// with coalesced global float atomics (see k_group8.h).
#pragma once
#include "k_group8.h"
#include "k_group8m.h"  // nlk_f4
#include "nlk_common.h"

constexpr float NLK_C12[12][12] = {
    {0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f, 0.288675129f},
    {0.404755682f, 0.377172232f, 0.323885143f, 0.248525813f, 0.156229854f, 0.0532870963f, -0.0532870963f, -0.156229854f, -0.248525813f, -0.323885143f, -0.377172232f, -0.404755682f},
    {0.394337565f, 0.288675129f, 0.105662435f, -0.105662435f, -0.288675129f, -0.394337565f, -0.394337565f, -0.288675129f, -0.105662435f, 0.105662435f, 0.288675129f, 0.394337565f},
    {0.377172232f, 0.156229854f, -0.156229854f, -0.377172232f, -0.377172232f, -0.156229854f, 0.156229854f, 0.377172232f, 0.377172232f, 0.156229854f, -0.156229854f, -0.377172232f},
    {0.353553385f, 0.f, -0.353553385f, -0.353553385f, 0.f, 0.353553385f, 0.353553385f, 0.f, -0.353553385f, -0.353553385f, 0.f, 0.353553385f},
    {0.323885143f, -0.156229854f, -0.404755682f, -0.0532870963f, 0.377172232f, 0.248525813f, -0.248525813f, -0.377172232f, 0.0532870963f, 0.404755682f, 0.156229854f, -0.323885143f},
    {0.288675129f, -0.288675129f, -0.288675129f, 0.288675129f, 0.288675129f, -0.288675129f, -0.288675129f, 0.288675129f, 0.288675129f, -0.288675129f, -0.288675129f, 0.288675129f},
    {0.248525813f, -0.377172232f, -0.0532870963f, 0.404755682f, -0.156229854f, -0.323885143f, 0.323885143f, 0.156229854f, -0.404755682f, 0.0532870963f, 0.377172232f, -0.248525813f},
    {0.204124153f, -0.408248305f, 0.204124153f, 0.204124153f, -0.408248305f, 0.204124153f, 0.204124153f, -0.408248305f, 0.204124153f, 0.204124153f, -0.408248305f, 0.204124153f},
    {0.156229854f, -0.377172232f, 0.377172232f, -0.156229854f, -0.156229854f, 0.377172232f, -0.377172232f, 0.156229854f, 0.156229854f, -0.377172232f, 0.377172232f, -0.156229854f},
    {0.105662435f, -0.288675129f, 0.394337565f, -0.394337565f, 0.288675129f, -0.105662435f, -0.105662435f, 0.288675129f, -0.394337565f, 0.394337565f, -0.288675129f, 0.105662435f},
    {0.0532870963f, -0.156229854f, 0.248525813f, -0.323885143f, 0.377172232f, -0.404755682f, 0.404755682f, -0.377172232f, 0.323885143f, -0.248525813f, 0.156229854f, -0.0532870963f},
};

// forward 1-D DCT-II of 12 registers (even/odd split: C[k][11-j] = (-1)^k C[k][j])
__device__ __forceinline__ void nlk_dct12_fwd(float (&p)[12]) {
  float s[6], d[6], y[12];
#pragma unroll
  for (int i = 0; i < 6; ++i) { s[i] = p[i] + p[11 - i]; d[i] = p[i] - p[11 - i]; }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float* z = (k & 1) ? d : s;
    float a = NLK_C12[k][0] * z[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) a = fmaf(NLK_C12[k][i], z[i], a);
    y[k] = a;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) p[k] = y[k];
}

__device__ __forceinline__ void nlk_dct12_inv(float (&y)[12]) {
  float E[6], O[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float e = NLK_C12[0][i] * y[0], o = NLK_C12[1][i] * y[1];
#pragma unroll
    for (int k = 2; k < 12; k += 2) { e = fmaf(NLK_C12[k][i], y[k], e); o = fmaf(NLK_C12[k + 1][i], y[k + 1], o); }
    E[i] = e; O[i] = o;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) { y[i] = E[i] + O[i]; y[11 - i] = E[i] - O[i]; }
}

// Column pass inside a DPP row of 16 lanes, four registers at a time:
//   p_i[u] <- ck[0]*p_i[u] + sum_{k=1..15} ck[k] * rot_k(p_i)[u]
// (ck[k] is zero wherever the rotated source lane is not one of the 12 rows).
#define NLK_ROR(k) "row_ror:" #k " row_mask:0xf bank_mask:0xf"
#define NLK_R4(k, c)                                              \
  "v_fmac_f32_dpp %0, %4, %" #c " " NLK_ROR(k) "\n\t"             \
  "v_fmac_f32_dpp %1, %5, %" #c " " NLK_ROR(k) "\n\t"             \
  "v_fmac_f32_dpp %2, %6, %" #c " " NLK_ROR(k) "\n\t"             \
  "v_fmac_f32_dpp %3, %7, %" #c " " NLK_ROR(k) "\n\t"
__device__ __forceinline__ void nlk_col12x4(float& p0, float& p1, float& p2, float& p3,
                                            const float (&ck)[16]) {
  float y0, y1, y2, y3;
  asm volatile(
      "s_nop 1\n\t"
      "v_mul_f32 %0, %8, %4\n\t"
      "v_mul_f32 %1, %8, %5\n\t"
      "v_mul_f32 %2, %8, %6\n\t"
      "v_mul_f32 %3, %8, %7\n\t"
      NLK_R4(1, 9) NLK_R4(2, 10) NLK_R4(3, 11) NLK_R4(4, 12) NLK_R4(5, 13) NLK_R4(6, 14)
      NLK_R4(7, 15) NLK_R4(8, 16) NLK_R4(9, 17) NLK_R4(10, 18) NLK_R4(11, 19) NLK_R4(12, 20)
      NLK_R4(13, 21) NLK_R4(14, 22) NLK_R4(15, 23)
      "s_nop 1"
      : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(ck[0]), "v"(ck[1]), "v"(ck[2]), "v"(ck[3]),
        "v"(ck[4]), "v"(ck[5]), "v"(ck[6]), "v"(ck[7]), "v"(ck[8]), "v"(ck[9]), "v"(ck[10]),
        "v"(ck[11]), "v"(ck[12]), "v"(ck[13]), "v"(ck[14]), "v"(ck[15]));
  p0 = y0; p1 = y1; p2 = y2; p3 = y3;
}

__device__ __forceinline__ void nlk_dct12x12_fwd(float (&p)[12], const float (&ck)[16]) {
  nlk_dct12_fwd(p);
  nlk_col12x4(p[0], p[1], p[2], p[3], ck);
  nlk_col12x4(p[4], p[5], p[6], p[7], ck);
  nlk_col12x4(p[8], p[9], p[10], p[11], ck);
}
__device__ __forceinline__ void nlk_dct12x12_inv(float (&p)[12], const float (&cik)[16]) {
  nlk_col12x4(p[0], p[1], p[2], p[3], cik);
  nlk_col12x4(p[4], p[5], p[6], p[7], cik);
  nlk_col12x4(p[8], p[9], p[10], p[11], cik);
  nlk_dct12_inv(p);
}

// Column pass through LDS (the default): the 15 dependent v_fmac_f32_dpp per register of the
// rotation scheme above were measured at HALF of the kernel's time (13.3 -> 6.4 ms at C3 with the
// column passes removed). Instead each lane writes its 12 row-transformed values as one row of a
// 12x12 scratch tile of its channel, reads back one COLUMN, and runs the same register transform
// along it: 84 VALU + 15 LDS instructions instead of 192 DPP ones. The coefficients then live
// transposed (lane = horizontal, register = vertical frequency); statistics, gains and shrinkage
// are elementwise and never notice, and the inverse transform undoes it: registers first, the
// transposition, registers again, which ends in the pixel layout (lane = row) the aggregation uses.
#ifndef NLK_G12_WAVES
#define NLK_G12_WAVES 3  // wavefronts per SIMD the register budget is cut for (168 VGPRs): the LDS round trips of
#endif                   // the transpositions want the third wavefront (9.2 -> 8.8 ms at C3; 4 would spill: 19 ms)
#define NLK_T12_CS 156  // floats per channel tile (12 x 12 + 12: the channels start 12 banks apart)
#define NLK_T12_FLOATS (4 * NLK_T12_CS)
__device__ __forceinline__ void nlk_transpose12(float (&p)[12], float* __restrict__ tile /* of this lane's channel */,
                                                int u, bool on) {
  if (on) {
    nlk_f4* row = (nlk_f4*)(tile + 12 * u);
    row[0] = nlk_f4{p[0], p[1], p[2], p[3]};
    row[1] = nlk_f4{p[4], p[5], p[6], p[7]};
    row[2] = nlk_f4{p[8], p[9], p[10], p[11]};
  }
  nlk_wave_lds_fence();
  if (on) {
#pragma unroll
    for (int j = 0; j < 12; ++j) p[j] = tile[12 * j + u];
  }
  nlk_wave_lds_fence();  // (the next transposition overwrites the tile)
}

__device__ __forceinline__ void nlk_load_row12(const float* __restrict__ p, float (&dst)[12]) {
  // (explicit global address space: see k_group8m.h)
  typedef const __attribute__((address_space(1))) nlk_f4u* gp4;
  const nlk_f4u a = *(gp4)(p);
  const nlk_f4u b = *(gp4)(p + 4);
  const nlk_f4u c = *(gp4)(p + 8);
  dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
  dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
  dst[8] = c.x; dst[9] = c.y; dst[10] = c.z; dst[11] = c.w;
}

template <int K>
__device__ __forceinline__ int nlk_ror_src(int lane) {  // source lane of row_ror:K for this lane
  return __builtin_amdgcn_update_dpp(0, lane, 0x120 + K, 0xF, 0xF, true);
}

template <int CH, bool SMO>
__global__ void __launch_bounds__(64, NLK_G12_WAVES)
k_group12(const float* __restrict__ img, const float* __restrict__ cur,
          const float* __restrict__ prev, const uint8_t* __restrict__ vmap, NlkGeom g,
          NlkGTile tl, const uint32_t* __restrict__ topk, const NlkTarget* __restrict__ tinfo,
          const uint32_t* __restrict__ gcoords, const uint8_t* __restrict__ active,
          const float* __restrict__ basis,   // [12][12] orthonormal DCT-II
          const float* __restrict__ window,  // [12][12] aggregation window
          float* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [(CH+1)][rh_max][rwp]
  constexpr int PSZ = 12, step = 6;
  const int lane = threadIdx.x;
  const int tile_id = nlk_xcd_tile(blockIdx.x, tl.ntx * tl.nty);
  if (tile_id >= tl.ntx * tl.nty) return;
  const int tile_x = tile_id % tl.ntx, tile_y = tile_id / tl.ntx;
  const int gx0 = tile_x * tl.tgx, gy0 = tile_y * tl.tgy;
  const int cx = min(tl.tgx, g.ngx - gx0), cy = min(tl.tgy, g.ngy - gy0);
  const int rx0 = max(gx0 * step - tl.wmax, 0);
  const int rx1 = min((gx0 + cx - 1) * step + tl.wmax + PSZ, g.w);
  const int ry0 = max(g.oy + gy0 * step - tl.wmax, 0);
  const int ry1 = min(g.oy + (gy0 + cy - 1) * step + tl.wmax + PSZ, g.h);
  const int rw = rx1 - rx0, rh = ry1 - ry0;
  const int rwp = tl.rwp, plane = rwp * tl.rh_max;
  for (int i = lane; i < (CH + 1) * plane; i += 64) smem[i] = 0.f;
  __syncthreads();

  // lane role
  const int u = lane & 15, c = lane >> 4;
  const bool lane_on = c < CH && u < PSZ;
  const int cc = c < CH ? c : 0, uu = u < PSZ ? u : 0;
  // coefficient of rotation k = basis entry (own row, source row) — found by rotating the lane index
  float ck[16], cik[16];
  {
    int src[16];
    src[0] = lane;
    src[1] = nlk_ror_src<1>(lane); src[2] = nlk_ror_src<2>(lane); src[3] = nlk_ror_src<3>(lane);
    src[4] = nlk_ror_src<4>(lane); src[5] = nlk_ror_src<5>(lane); src[6] = nlk_ror_src<6>(lane);
    src[7] = nlk_ror_src<7>(lane); src[8] = nlk_ror_src<8>(lane); src[9] = nlk_ror_src<9>(lane);
    src[10] = nlk_ror_src<10>(lane); src[11] = nlk_ror_src<11>(lane); src[12] = nlk_ror_src<12>(lane);
    src[13] = nlk_ror_src<13>(lane); src[14] = nlk_ror_src<14>(lane); src[15] = nlk_ror_src<15>(lane);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int su = src[k] & 15;
      const bool ok = u < PSZ && su < PSZ;
      ck[k] = ok ? basis[u * PSZ + su] : 0.f;    // forward: Y[u] = sum C[u][y] T[y]
      cik[k] = ok ? basis[su * PSZ + u] : 0.f;   // inverse: x[y=u] = sum C[u'][y] Y[u']
    }
  }
#ifndef NLK_GROUP12_DPP
  float* const t12 = smem + (((CH + 1) * plane + 3) & ~3) + cc * NLK_T12_CS;
#define NLK_FWD12(p) do { nlk_dct12_fwd(p); nlk_transpose12(p, t12, uu, lane_on); nlk_dct12_fwd(p); } while (0)
#define NLK_INV12(p) do { nlk_dct12_inv(p); nlk_transpose12(p, t12, uu, lane_on); nlk_dct12_inv(p); } while (0)
#else
#define NLK_FWD12(p) nlk_dct12x12_fwd(p, ck)
#define NLK_INV12(p) nlk_dct12x12_inv(p, cik)
#endif
  float wrow[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) wrow[k] = window[uu * PSZ + k];
  const size_t npix = (size_t)g.w * g.h;
  const float* img_c = img + cc * npix + (size_t)uu * g.w;
  const float* cur_c = cur + cc * npix + (size_t)uu * g.w;
  const float* prev_c = prev ? prev + cc * npix + (size_t)uu * g.w : img_c;
  const float* src_c = g.have_basic ? cur_c : img_c;  // patches that get filtered
  const float s2 = g.sigma2;

  int rec_act = 0, rec_nsel = 0, rec_nagg = 0, rec_flags = 0;
  if (lane < cx * cy) {
    const int ty = lane / cx, tx = lane - ty * cx;
    const size_t t = (size_t)(gy0 + ty) * g.ngx + gx0 + tx;
    rec_act = active[t];
    const NlkTarget info = tinfo[t];
    rec_nsel = info.nsel; rec_nagg = info.nagg; rec_flags = info.flags;
  }

  for (int tt = 0; tt < cx * cy; ++tt) {
    if (!__builtin_amdgcn_readlane(rec_act, tt)) continue;
    const int nagg = __builtin_amdgcn_readlane(rec_nagg, tt);
    if (nagg == 0) continue;
    const int ty = tt / cx, tx = tt - ty * cx;
    const size_t t = (size_t)(gy0 + ty) * g.ngx + gx0 + tx;
    const bool prev_p = __builtin_amdgcn_readlane(rec_flags, tt) & 1;
    const int k = __builtin_amdgcn_readlane(rec_nsel, tt);

    uint32_t qreg[2], greg[2];
    uint64_t vbits[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int i = lane + 64 * m;
      qreg[m] = i < k ? topk[t * g.kmax + i] : 0u;
      greg[m] = i < nagg ? gcoords[t * g.gstride + i] : 0u;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int i = lane + 64 * m;
      const int org = nlk_y(qreg[m]) * g.w + nlk_x(qreg[m]);
      vbits[m] = __ballot(prev_p && i < k && vmap[org]);
    }
    auto cand = [&](int i) -> uint32_t {
      return i < 64 ? __builtin_amdgcn_readlane(qreg[0], i) : __builtin_amdgcn_readlane(qreg[1], i - 64);
    };
    auto member = [&](int i) -> uint32_t {
      return i < 64 ? __builtin_amdgcn_readlane(greg[0], i) : __builtin_amdgcn_readlane(greg[1], i - 64);
    };

    // ---------------- pass A: statistics over the k kept candidates
    float m1[12], v1[12], mb[12], vb[12], v01[12], m0[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) m1[r] = v1[r] = mb[r] = vb[r] = v01[r] = m0[r] = 0.f;
    int np0 = 0, np1 = 0;
    // The rows of the next candidate are requested before the current one is transformed.
    // All loads are unconditional (idle lanes read row 0 of channel 0, a candidate without a
    // valid previous patch reads the image instead: finite data that is never used), so the
    // waits stay partial.
    float a[12], b[12], na[12], nb[12];
    const ptrdiff_t prev_off = prev_c - img_c;
    auto rows_of = [&](int i, float (&ra)[12], float (&rb)[12]) {
      const uint32_t q = cand(i);
      const int org = nlk_y(q) * g.w + nlk_x(q);
      const bool v = (vbits[i >> 6] >> (i & 63)) & 1ull;
      nlk_load_row12(img_c + org, ra);
      nlk_load_row12(img_c + (v ? prev_off : (ptrdiff_t)0) + org, rb);
    };
    if (k > 0) rows_of(0, na, nb);
    for (int i = 0; i < k; ++i) {
      const bool v = (vbits[i >> 6] >> (i & 63)) & 1ull;
#pragma unroll
      for (int r = 0; r < 12; ++r) { a[r] = na[r]; b[r] = nb[r]; }
      rows_of(i + 1 < k ? i + 1 : i, na, nb);
      NLK_FWD12(a);
      if (v) NLK_FWD12(b);
      np1++;
      if (v) np0++;
      const float inp1 = __builtin_amdgcn_rcpf((float)np1);
      const float inp0 = v ? __builtin_amdgcn_rcpf((float)np0) : 0.f;
      const bool in_group = v && np0 <= g.ntagg;
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        const float d = a[r] - m1[r];
        m1[r] = fmaf(d, inp1, m1[r]);
        v1[r] = fmaf(d, a[r] - m1[r], v1[r]);
      }
      if (v) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          const float d = b[r] - mb[r];
          mb[r] = fmaf(d, inp0, mb[r]);
          vb[r] = fmaf(d, b[r] - mb[r], vb[r]);
          const float tr = b[r] - a[r];
          v01[r] = fmaf(tr, tr, v01[r]);
        }
        if (!SMO && in_group) {
#pragma unroll
          for (int r = 0; r < 12; ++r) m0[r] = fmaf(b[r] - m0[r], inp0, m0[r]);
        }
      }
    }

    // ---------------- gains (reference: :799-811, :859-904; smoother :1683-1776)
    float gain[12], mu[12];
    float part_sum = 0.f;
    {
      const float in1 = np1 ? 1.f / (float)np1 : 0.f;
      const float in0 = np0 ? 1.f / (float)np0 : 0.f;
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        const float V1 = v1[r] * in1, V0 = vb[r] * in0, V01 = v01[r] * in0;
        float ga, term, m;
        if (SMO) {
          ga = V1 / (V1 + g.beta_t * V01);
          const float pv = V0 - g.beta_t * V01;
          term = (1 - ga * ga) * V1 + ga * ga * (pv > 0.f ? pv : 0.f);
          m = 0.f;
        } else if (np0 > 0) {
          const float d = V01 - (g.have_basic ? 0.f : s2);
          const float vv = V0 + (0.f > d ? 0.f : d);
          ga = vv / (vv + g.beta_t * s2);
          term = (1 - ga * ga) * vv + ga * ga * s2;
          m = m0[r];
        } else {
          const float d = V1 - (g.have_basic ? 0.f : s2);
          const float vv = 0.f > d ? 0.f : d;
          ga = vv / (vv + g.beta_x * s2);
          term = ga * vv;
          m = m1[r];
        }
        // idle lanes (u >= 12) share the DPP row: keep them finite (0 * NaN would leak)
        gain[r] = lane_on ? ga : 0.f;
        mu[r] = lane_on ? m : 0.f;
        if (lane_on) part_sum += term;
      }
    }
    float vp = nlk_wave_sum8(part_sum) * (float)nagg;
    const bool passthrough = SMO && np0 == 0;  // reference: :1795-1804
    if (passthrough) vp = 0.f;
    const float wgt = 1.f / (vp > 1e-6f ? vp : 1e-6f);

    // ---------------- pass B: shrink, invert and aggregate the group members
    auto add_patch = [&](int qx, int qy, const float (&px)[12]) {
      const int lx = qx - rx0, ly = qy - ry0;
      if (lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh) {
        float* dst = smem + cc * plane + (ly + uu) * rwp + lx;
        float* dw = smem + CH * plane + (ly + uu) * rwp + lx;
        float o[12], ow[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) { o[r] = dst[r]; if (c == 0) ow[r] = dw[r]; }
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          const float ww = wgt * wrow[r];
          dst[r] = o[r] + ww * px[r];
          if (c == 0) dw[r] = ow[r] + ww;
        }
      } else {
        float* dst = acc + (size_t)cc * npix + (size_t)(qy + uu) * g.w + qx;
        float* dw = acc + (size_t)CH * npix + (size_t)(qy + uu) * g.w + qx;
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          const float ww = wgt * wrow[r];
          unsafeAtomicAdd(dst + r, ww * px[r]);
          if (c == 0) unsafeAtomicAdd(dw + r, ww);
        }
      }
    };
    if (!SMO) {
      for (int n0 = 0; n0 < nagg; n0 += 2) {  // member n0 in a[], member n0+1 in b[]
        const bool two = n0 + 1 < nagg;
        const uint32_t qa = member(n0), qb = member(two ? n0 + 1 : n0);
        if (lane_on) nlk_load_row12(src_c + nlk_y(qa) * g.w + nlk_x(qa), a);
        if (lane_on && two) nlk_load_row12(src_c + nlk_y(qb) * g.w + nlk_x(qb), b);
        NLK_FWD12(a);
        if (two) NLK_FWD12(b);
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          a[r] = gain[r] * a[r] + (1 - gain[r]) * mu[r];
          b[r] = gain[r] * b[r] + (1 - gain[r]) * mu[r];
        }
        NLK_INV12(a);
        if (two) NLK_INV12(b);
        if (lane_on) add_patch(nlk_x(qa), nlk_y(qa), a);
        if (lane_on && two) add_patch(nlk_x(qb), nlk_y(qb), b);
      }
    } else {
      for (int n = 0; n < nagg; ++n) {
        const uint32_t q = member(n);
        const int qx = nlk_x(q), qy = nlk_y(q);
        if (lane_on) nlk_load_row12(src_c + qy * g.w + qx, a);
        if (!passthrough) {
          if (lane_on) nlk_load_row12(prev_c + qy * g.w + qx, b);
          NLK_FWD12(a);
          NLK_FWD12(b);
#pragma unroll
          for (int r = 0; r < 12; ++r) a[r] = (1 - gain[r]) * a[r] + gain[r] * b[r];  // reference: :1775
          NLK_INV12(a);
        }
        if (lane_on) add_patch(qx, qy, a);
      }
    }
  }

  // ---------------- flush the tile accumulator (coalesced rows, skip untouched)
  __syncthreads();
  for (int p = 0; p <= CH; ++p)
    for (int y = 0; y < rh; ++y) {
      const float* srow = smem + p * plane + y * rwp;
      float* drow = acc + (size_t)p * npix + (size_t)(ry0 + y) * g.w + rx0;
      for (int xx = lane; xx < rw; xx += 64) {
        const float v = srow[xx];
        if (v != 0.f) unsafeAtomicAdd(drow + xx, v);
      }
    }
}

#undef NLK_FWD12
#undef NLK_INV12
