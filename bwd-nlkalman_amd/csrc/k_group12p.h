// k_group12p.h — group processing for 12x12 patches (BASELINE config C3), packed lanes
// (reference: src/nlkalman.c:713-932 filter, :1603-1845 smoother).
//
// The separable 12x12 DCT is run as 1-D passes on 12 registers of a lane (k_dct12.h: a 66 / 68
// operation flow graph, no cross-lane traffic), a 12x12 transposition through LDS, and the same
// pass again. What this kernel changes against k_group12.h (kept as the comparison variant,
// NLK_GROUP12_ROWS=1), where lane = (channel, row) left 28 of 64 lanes idle:
//
//   lane = 12 * slot + u : row u of patch slot 0..4 (60 lanes busy). One channel at a time,
//   * pass A: a step transforms FIVE candidates (image patch row in a[], previous-frame patch row
//     in b[] of the same lane), so the statistics of a coefficient are sums over the steps in
//     registers plus one reduction over the 5 slots per channel (through LDS, one statistic at a
//     time; lanes of slots 0..2 then own 4 coefficients each and compute their gains);
//     sums are taken around x0 = the coefficient of the first candidate (as k_group8m.h). A slot
//     without a candidate re-reads candidate 0 and a candidate without a valid previous patch
//     reads candidate 0's image as "previous": bit-identical arithmetic gives exact zeros, so only
//     the transition term needs a mask;
//   * pass B: a step shrinks and inverts TEN group members (filter) or five (smoother: image and
//     previous patch of a member in one lane); the pixel rows are then staged in LDS slot by slot
//     and added to the private accumulator tile by 48 lanes = 12 rows x 4 blocks of 3 pixels of ONE
//     member, so that no two lanes of an instruction touch the same tile entry.
// A workgroup (one wavefront) owns ONE target. Its accumulator tile holds two planes only: the
// weights and the values of the channel in flight (pass B is channel-major; the plane is flushed to
// HBM with float atomics and cleared after each channel), so that tile + transposition scratch +
// gains stay at 11 KB and the register budget, not LDS, sets the occupancy (3 wavefronts per SIMD;
// the f32 vector ALU needs them: 4.2 cycles per instruction at 2, 3.0 at 3, profiles/README.md).
// Coefficients live transposed between the passes exactly as in k_group12.h (statistics, gains and
// shrinkage are elementwise).
//
// The matrix cores are not used here on purpose: see k_dct12.h for the arithmetic.
#pragma once
#include "k_dct12.h"
#include "k_group8.h"
#include "k_group8m.h"
#include "nlk_common.h"

#ifndef NLK_G12P_WAVES
#define NLK_G12P_WAVES 3  // wavefronts per SIMD the register budget is cut for (168)
#endif
// ordering of this wavefront's LDS writes and reads: compiler-only (nlk_common.h); NLK_P12_WAITS
// builds the variant that also waits for the LDS counter (5.22 ms against 5.07 at C3)
#ifdef NLK_P12_WAITS
#define NLK_P12_SYNC nlk_wave_lds_fence
#else
#define NLK_P12_SYNC nlk_wave_lds_order
#endif
#define NLK_P12_TS 172                    // floats per slot of the transposition scratch (= 12 mod 32: the 5 slots' columns sit on disjoint banks)
#define NLK_P12_SCRATCH (5 * NLK_P12_TS)  // also holds [5][12][12] reduction partials / staged pixel rows
#define NLK_P12_GAINS(CH) ((CH) * 2 * 144)

// forward / inverse 2-D transform of the lane's row set: registers, transposition, registers
__device__ __forceinline__ void nlk_p12_transpose(float (&p)[12], float* __restrict__ tile, int u, bool on) {
  if (on) {
    nlk_f4* row = (nlk_f4*)(tile + 12 * u);
    row[0] = nlk_f4{p[0], p[1], p[2], p[3]};
    row[1] = nlk_f4{p[4], p[5], p[6], p[7]};
    row[2] = nlk_f4{p[8], p[9], p[10], p[11]};
  }
  NLK_P12_SYNC();
  if (on) {
#pragma unroll
    for (int j = 0; j < 12; ++j) p[j] = tile[12 * j + u];
  }
  NLK_P12_SYNC();
}

__device__ __forceinline__ void nlk_p12_load_row(const float* __restrict__ p, float (&dst)[12]) {
  typedef const __attribute__((address_space(1))) nlk_f4u* gp4;  // (global, not flat: see k_group8m.h)
  const nlk_f4u a = *(gp4)(p);
  const nlk_f4u b = *(gp4)(p + 4);
  const nlk_f4u c = *(gp4)(p + 8);
  dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
  dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
  dst[8] = c.x; dst[9] = c.y; dst[10] = c.z; dst[11] = c.w;
}

template <int CH, bool SMO>
__global__ void __launch_bounds__(64, NLK_G12P_WAVES)
k_group12p(const float* __restrict__ img, const float* __restrict__ cur,
           const float* __restrict__ prev, const uint8_t* __restrict__ vmap, NlkGeom g,
           NlkGTile tl, const uint32_t* __restrict__ topk, const NlkTarget* __restrict__ tinfo,
           const uint32_t* __restrict__ gcoords, const uint8_t* __restrict__ active,
           const float* __restrict__ basis,   // unused: the transform is the flow graph of k_dct12.h
           const float* __restrict__ window,  // [12][12] aggregation window
           float* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // values | weights | scratch | gains
  constexpr int PSZ = 12, step = 6;
  const int lane = threadIdx.x;
  const int ngrid = g.ngx * g.ngy;
  const int ti = nlk_xcd_tile(blockIdx.x, ngrid);  // one target per workgroup (tl.tgx = tl.tgy = 1)
  if (ti >= ngrid) return;
  if (!active[ti]) return;
  const NlkTarget info = tinfo[ti];
  const int nagg = info.nagg, k = info.nsel;
  if (nagg == 0) return;
  const size_t t = (size_t)ti;
  const int gy = ti / g.ngx, gx = ti - gy * g.ngx;
  const int rx0 = max(gx * step - tl.wmax, 0);
  const int rx1 = min(gx * step + tl.wmax + PSZ, g.w);
  const int ry0 = max(g.oy + gy * step - tl.wmax, 0);
  const int ry1 = min(g.oy + gy * step + tl.wmax + PSZ, g.h);
  const int rw = rx1 - rx0, rh = ry1 - ry0;
  const int rwp = tl.rwp, plane = tl.plane;
  for (int i = lane; i < 2 * plane / 4; i += 64)  // (plane is a multiple of 4)
    reinterpret_cast<nlk_f4*>(smem)[i] = nlk_f4{0.f, 0.f, 0.f, 0.f};
  float* const vplane = smem;           // values of the channel in flight
  float* const wplane = smem + plane;   // weights
  float* const scratch = smem + 2 * plane;
  float* const gbuf = scratch + NLK_P12_SCRATCH;  // [CH][gain | (1-gain)*mean][u][12]

  // lane role: slot (0..4; 5 = the four idle lanes) and row / column u
  const int slot = lane / 12, u = lane - 12 * slot;
  const bool on = slot < 5;
  const int sl = on ? slot : 0;
  float* const tsc = scratch + sl * NLK_P12_TS;
#define NLK_P12_FWD(p) do { nlk_dct12_fast_fwd(p); nlk_p12_transpose(p, tsc, u, on); nlk_dct12_fast_fwd(p); } while (0)
#define NLK_P12_INV(p) do { nlk_dct12_fast_inv(p); nlk_p12_transpose(p, tsc, u, on); nlk_dct12_fast_inv(p); } while (0)
  // aggregation role: row u, pixels 3*slot .. 3*slot+2 of ONE member (slots 0..3: 48 lanes)
  const bool agg_on = slot < 4;
  float w3[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) w3[e] = window[u * PSZ + 3 * (agg_on ? slot : 0) + e];
  const size_t npix = (size_t)g.w * g.h;
  const float* src = g.have_basic ? cur : img;  // patches that get filtered
  const ptrdiff_t prev_off = prev ? prev - img : 0;
  const ptrdiff_t src_off = src - img;
  const float s2 = g.sigma2;

  uint32_t qreg[2], greg[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int i = lane + 64 * m;
    qreg[m] = i < k ? topk[t * g.kmax + i] : 0u;
    greg[m] = i < nagg ? gcoords[t * g.gstride + i] : 0u;
  }
  const uint64_t vbits[2] = {info.vbits[0], info.vbits[1]};
  const int np0a = __popcll(vbits[0]);
  const int np0 = np0a + __popcll(vbits[1]), np1 = k;
  const int ngrp = min(np0, g.ntagg);
  const float in1 = np1 ? 1.f / (float)np1 : 0.f;
  const float in0 = np0 ? 1.f / (float)np0 : 0.f;
  const float ing = ngrp ? 1.f / (float)ngrp : 0.f;
  const bool passthrough = SMO && np0 == 0;  // reference: :1795-1804
  const bool hp = np0 > 0;                   // previous-frame statistics exist (Kalman branch / smoother)
  // the step in which the last group member (the ngrp-th candidate with a valid previous patch) is
  // transformed: the group mean of the previous-frame coefficients is reduced there
  int it_m = -1;
  if (!SMO && hp) {
    uint64_t lo = vbits[0], hi = vbits[1];
    for (int n = 1; n < ngrp; ++n) {
      if (lo) lo &= lo - 1; else hi &= hi - 1;
    }
    const int c_last = lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll(hi);
    it_m = c_last / 5;
  }
  // (entry i of a list held one per lane in two registers; the lanes of a step may sit on both
  // sides of entry 64, and a bpermute delivers the SOURCE lane's operand: select after it. Always
  // called by all lanes: a bpermute under divergence reads 0 from the masked-off lanes.)
  auto list_at = [&](const uint32_t (&reg)[2], int i, int n) -> uint32_t {
    uint32_t q = nlk_bperm_u(reg[0], i & 63);
    if (n > 64) {
      const uint32_t q1 = nlk_bperm_u(reg[1], i & 63);
      q = i < 64 ? q : q1;
    }
    return q;
  };
  auto cand_org = [&](int i) -> int {  // origin (row u of channel 0) of candidate i
    const uint32_t q = list_at(qreg, i, k);
    return (nlk_y(q) + u) * g.w + nlk_x(q);
  };
  auto memb_org = [&](int i) -> int {
    const uint32_t q = list_at(greg, i, nagg);
    return (nlk_y(q) + u) * g.w + nlk_x(q);
  };
  __syncthreads();

  // ---------------- pass A: statistics over the k kept candidates, one channel at a time
  float part_sum = 0.f;
  if (!passthrough) {
    const int nit = (k + 4) / 5;
    const int org0 = cand_org(0);
    const bool owner = slot < 3;  // lane (slot < 3, u) owns coefficients (u, 4*slot .. 4*slot+3)
    float* const red = scratch + (sl * 12 + u) * 12;                       // partials as [slot][u][12]
    const float* const rd = scratch + u * 12 + 4 * (owner ? slot : 0);
    auto put12 = [&](const float (&v)[12]) {
      if (on) {
        ((nlk_f4*)red)[0] = nlk_f4{v[0], v[1], v[2], v[3]};
        ((nlk_f4*)red)[1] = nlk_f4{v[4], v[5], v[6], v[7]};
        ((nlk_f4*)red)[2] = nlk_f4{v[8], v[9], v[10], v[11]};
      }
      NLK_P12_SYNC();
    };
    auto sum5 = [&]() -> nlk_f4 {  // sum over the 5 slots of the owned coefficients
      nlk_f4 tsum = nlk_f4{0.f, 0.f, 0.f, 0.f};
      if (owner) {
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) tsum += *(const nlk_f4*)(rd + s5 * 144);
      }
      NLK_P12_SYNC();
      return tsum;
    };
    // two copies of the channel loop (with / without previous-frame statistics): with `hp` a
    // run-time condition inside one loop the register allocation spilled 130 registers
    auto pass_a = [&](auto has_prev) {
     constexpr bool HP = decltype(has_prev)::value;
     for (int c = 0; c < CH; ++c) {
      const float* img_c = img + c * npix;
      // S0/S1 image, S2/S3 previous frame, S4 squared image-previous difference: sums of
      // deviations from x0 (see the header)
      float S[HP ? 5 : 2][12], x0[12];
#pragma unroll
      for (int a = 0; a < (HP ? 5 : 2); ++a)
#pragma unroll
        for (int r = 0; r < 12; ++r) S[a][r] = 0.f;
      nlk_f4 tot5 = nlk_f4{0.f, 0.f, 0.f, 0.f};  // previous frame over the group members (owners)
      // the slot's candidate in step `it`: validity mask and the addresses of its two rows
      auto job = [&](int it, float& vm, const float*& pa, const float*& pb) {
        const int ci = 5 * it + sl;
        const bool valid = on && ci < k;
        const int cl = valid ? ci : 0;
        const uint64_t vw = cl < 64 ? vbits[0] : vbits[1];
        const bool v = valid && ((vw >> (cl & 63)) & 1ull);
        vm = v ? 1.f : 0.f;
        const int org = cand_org(cl);
        pa = img_c + org;
        pb = v ? img_c + prev_off + org : img_c + org0;  // (no valid previous patch: candidate 0's image = x0)
      };
      // one step: the rows in (a, b) are transformed and accumulated while the next step's rows
      // travel into (na, nb); the two register sets swap roles from step to step
      auto stage = [&](int it, float vm, float (&a)[12], float (&b)[12], float& nvm, float (&na)[12],
                       float (&nb)[12]) {
        const float *pa, *pb;
        job(it + 1 < nit ? it + 1 : it, nvm, pa, pb);
        nlk_p12_load_row(pa, na);
        if (HP) nlk_p12_load_row(pb, nb);
        NLK_P12_FWD(a);
        if (HP) NLK_P12_FWD(b);
        if (it == 0) {
#pragma unroll
          for (int r = 0; r < 12; ++r) x0[r] = nlk_bperm(a[r], u);  // slot 0 holds candidate 0
        }
        if (HP && !SMO && it == it_m) {
          // group mean: previous-frame deviations of the members so far + this step's members
          const int ci = 5 * it + sl;
          const uint64_t vw = ci < 64 ? vbits[0] : vbits[1];
          const int rank = (ci < 64 ? 0 : np0a) + __popcll(vw & ((1ull << (ci & 63)) - 1ull));
          const float gm = (vm != 0.f && rank < g.ntagg) ? 1.f : 0.f;
          if (on) {  // (four values at a time: twelve more live registers would not fit)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              ((nlk_f4*)red)[j] = nlk_f4{fmaf(gm, b[4 * j] - x0[4 * j], S[2][4 * j]),
                                         fmaf(gm, b[4 * j + 1] - x0[4 * j + 1], S[2][4 * j + 1]),
                                         fmaf(gm, b[4 * j + 2] - x0[4 * j + 2], S[2][4 * j + 2]),
                                         fmaf(gm, b[4 * j + 3] - x0[4 * j + 3], S[2][4 * j + 3])};
          }
          NLK_P12_SYNC();
          tot5 = sum5();
        }
        if constexpr (HP) {
#pragma unroll
          for (int r = 0; r < 12; ++r) {
            const float da = a[r] - x0[r], db = b[r] - x0[r];
            S[0][r] += da;
            S[1][r] = fmaf(da, da, S[1][r]);
            S[2][r] += db;
            S[3][r] = fmaf(db, db, S[3][r]);
            const float df = db - da;  // reference: :769-783, smoother :1659-1667
            S[4][r] = fmaf(vm * df, df, S[4][r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 12; ++r) {
            const float da = a[r] - x0[r];
            S[0][r] += da;
            S[1][r] = fmaf(da, da, S[1][r]);
          }
        }
      };
      float A1[12], B1[12], A2[12], B2[12], vm1, vm2 = 0.f;
#pragma unroll
      for (int r = 0; r < 12; ++r) B1[r] = B2[r] = 0.f;
      {
        const float *pa, *pb;
        job(0, vm1, pa, pb);
        nlk_p12_load_row(pa, A1);
        if (HP) nlk_p12_load_row(pb, B1);
      }
      for (int it = 0; it < nit; it += 2) {
        stage(it, vm1, A1, B1, vm2, A2, B2);
        if (it + 1 < nit) stage(it + 1, vm2, A2, B2, vm1, A1, B1);
      }
      // ---- sums over the 5 slots, one statistic at a time
      nlk_f4 tot[5], x04;
      put12(x0);
      x04 = owner ? *(const nlk_f4*)rd : nlk_f4{0.f, 0.f, 0.f, 0.f};  // (every slot holds the same x0)
      NLK_P12_SYNC();
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        tot[st] = nlk_f4{0.f, 0.f, 0.f, 0.f};
        if (st < (HP ? 5 : 2)) {
          put12(S[st < (HP ? 5 : 2) ? st : 0]);
          tot[st] = sum5();
        }
      }
      // ---- gains of the owned coefficients (reference: :799-811, :859-904; smoother :1683-1776)
      if (owner) {
        nlk_f4 ga4, mu4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float mean1 = x04[j] + tot[0][j] * in1;
          const float v1 = (tot[1][j] - tot[0][j] * tot[0][j] * in1) * in1;  // image variance
          const float v0 = (tot[3][j] - tot[2][j] * tot[2][j] * in0) * in0;  // previous-frame variance
          const float v01n = tot[4][j] * in0;
          float ga, term, m;
          if (SMO) {
            ga = v1 / (v1 + g.beta_t * v01n);
            const float pv = v0 - g.beta_t * v01n;
            term = (1 - ga * ga) * v1 + ga * ga * (pv > 0.f ? pv : 0.f);
            m = 0.f;
          } else if (HP) {
            const float d = v01n - (g.have_basic ? 0.f : s2);
            const float v = v0 + (0.f > d ? 0.f : d);
            ga = v / (v + g.beta_t * s2);
            term = (1 - ga * ga) * v + ga * ga * s2;
            m = x04[j] + tot5[j] * ing;
          } else {
            const float d = v1 - (g.have_basic ? 0.f : s2);
            const float v = 0.f > d ? 0.f : d;
            ga = v / (v + g.beta_x * s2);
            term = ga * v;
            m = mean1;
          }
          part_sum += term;
          ga4[j] = ga;
          mu4[j] = (1 - ga) * m;  // filter: a*PG + (1-a)*M (reference: :879, :902)
        }
        *(nlk_f4*)(gbuf + ((c * 2 + 0) * 12 + u) * 12 + 4 * slot) = ga4;
        *(nlk_f4*)(gbuf + ((c * 2 + 1) * 12 + u) * 12 + 4 * slot) = mu4;
      }
      NLK_P12_SYNC();
     }
    };
    if (hp) pass_a(std::true_type{});
    else pass_a(std::false_type{});
  }
  // the reference adds the same per-coefficient terms once per group member
  float vp = nlk_wave_sum8(part_sum) * (float)nagg;
  if (passthrough) vp = 0.f;
  const float wgt = 1.f / (vp > 1e-6f ? vp : 1e-6f);
  float ww[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) ww[e] = wgt * w3[e];

  // ---------------- pass B: shrink, invert and aggregate the group members
  // rows staged in `scratch` as [slot][u][12]; member n0 + s of the round is added by 48 lanes
  auto add_round = [&](float (&px)[12], int n0, int c) {
    if (on) {
      nlk_f4* row = (nlk_f4*)(scratch + (sl * 12 + u) * 12);
      row[0] = nlk_f4{px[0], px[1], px[2], px[3]};
      row[1] = nlk_f4{px[4], px[5], px[6], px[7]};
      row[2] = nlk_f4{px[8], px[9], px[10], px[11]};
    }
    NLK_P12_SYNC();
#pragma unroll 1
    for (int s = 0; s < 5; ++s) {
      const int mi = n0 + s;
      if (mi >= nagg) break;
      const uint32_t q = mi < 64 ? __builtin_amdgcn_readlane(greg[0], mi)
                                 : __builtin_amdgcn_readlane(greg[1], mi - 64);
      const int qx = nlk_x(q), qy = nlk_y(q);
      const int lx = qx - rx0, ly = qy - ry0;
      if (agg_on) {
        const float* sp = scratch + (s * 12 + u) * 12 + 3 * slot;
        const float v0 = sp[0], v1 = sp[1], v2 = sp[2];
        if (lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh) {
          float* dst = vplane + (ly + u) * rwp + lx + 3 * slot;
          const float o0 = dst[0], o1 = dst[1], o2 = dst[2];
          dst[0] = fmaf(ww[0], v0, o0); dst[1] = fmaf(ww[1], v1, o1); dst[2] = fmaf(ww[2], v2, o2);
          if (c == 0) {
            float* dw = wplane + (ly + u) * rwp + lx + 3 * slot;
            const float p0 = dw[0], p1 = dw[1], p2 = dw[2];
            dw[0] = p0 + ww[0]; dw[1] = p1 + ww[1]; dw[2] = p2 + ww[2];
          }
        } else {
          float* dst = acc + (size_t)c * npix + (size_t)(qy + u) * g.w + qx + 3 * slot;
          unsafeAtomicAdd(dst + 0, ww[0] * v0); unsafeAtomicAdd(dst + 1, ww[1] * v1); unsafeAtomicAdd(dst + 2, ww[2] * v2);
          if (c == 0) {
            float* dw = acc + (size_t)CH * npix + (size_t)(qy + u) * g.w + qx + 3 * slot;
            unsafeAtomicAdd(dw + 0, ww[0]); unsafeAtomicAdd(dw + 1, ww[1]); unsafeAtomicAdd(dw + 2, ww[2]);
          }
        }
      }
    }
    NLK_P12_SYNC();
  };
  // a tile plane -> HBM (coalesced rows, untouched entries skipped), cleared for the next channel
  const bool two_rows = rw <= 32;  // a narrow tile puts two rows on the 64 lanes
  const int fx = two_rows ? (lane & 31) : lane, fy = two_rows ? (lane >> 5) : 0;
  const int sx = two_rows ? 32 : 64, sy = two_rows ? 2 : 1;
  auto flush = [&](float* sp, int p, bool clear) {
    float* dp = acc + (size_t)p * npix + (size_t)ry0 * g.w + rx0;
#pragma unroll 4
    for (int y = fy; y < rh; y += sy)
      for (int xx = fx; xx < rw; xx += sx) {
        const float v = sp[y * rwp + xx];
        if (v != 0.f) {
          unsafeAtomicAdd(dp + (size_t)y * g.w + xx, v);
          if (clear) sp[y * rwp + xx] = 0.f;
        }
      }
    NLK_P12_SYNC();
  };
  for (int c = 0; c < CH; ++c) {
    const float* img_c = img + c * npix;
    float gain[12], mu[12], a[12], b[12];
    if (!passthrough) {
      const nlk_f4* gp = (const nlk_f4*)(gbuf + ((c * 2 + 0) * 12 + u) * 12);
      const nlk_f4* mp = (const nlk_f4*)(gbuf + ((c * 2 + 1) * 12 + u) * 12);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const nlk_f4 gv = gp[j], mv = mp[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gain[4 * j + e] = gv[e]; mu[4 * j + e] = mv[e]; }
      }
    }
    if (passthrough) {
      nlk_p12_load_row(img_c + src_off + memb_org(0), a);
      add_round(a, 0, c);
    } else if (!SMO) {
      for (int n0 = 0; n0 < nagg; n0 += 10) {  // member n0 + slot in a[], member n0 + 5 + slot in b[]
        const int ma = min(n0 + sl, nagg - 1), mb = min(n0 + 5 + sl, nagg - 1);
        const bool two = n0 + 5 < nagg;
        nlk_p12_load_row(img_c + src_off + memb_org(ma), a);
        if (two) nlk_p12_load_row(img_c + src_off + memb_org(mb), b);
        NLK_P12_FWD(a);
        if (two) NLK_P12_FWD(b);
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          a[r] = fmaf(gain[r], a[r], mu[r]);
          b[r] = fmaf(gain[r], b[r], mu[r]);
        }
        NLK_P12_INV(a);
        if (two) NLK_P12_INV(b);
        add_round(a, n0, c);
        if (two) add_round(b, n0 + 5, c);
      }
    } else {
      for (int n0 = 0; n0 < nagg; n0 += 5) {  // image and previous-frame patch of member n0 + slot
        const int ma = min(n0 + sl, nagg - 1);
        const int org = memb_org(ma);
        nlk_p12_load_row(img_c + src_off + org, a);
        nlk_p12_load_row(img_c + prev_off + org, b);
        NLK_P12_FWD(a);
        NLK_P12_FWD(b);
#pragma unroll
        for (int r = 0; r < 12; ++r) a[r] = (1 - gain[r]) * a[r] + gain[r] * b[r];  // reference: :1775
        NLK_P12_INV(a);
        add_round(a, n0, c);
      }
    }
    flush(vplane, c, c + 1 < CH);
  }
  flush(wplane, CH, false);
}

#undef NLK_P12_FWD
#undef NLK_P12_INV
