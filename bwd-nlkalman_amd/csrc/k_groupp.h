// k_groupp.h — group processing with PACKED LANES, any patch size 2..16 and any channel count
// (reference: src/nlkalman.c:713-932 filter, :1603-1845 smoother).
//
// The separable PSZ x PSZ DCT runs as 1-D passes on PSZ registers of a lane (k_dct8.h / k_dct12.h:
// flow graphs of 36 / 66 operations; other sizes: the basis matrix, folded by its even/odd
// symmetry where PSZ is even), a PSZ x PSZ transposition through LDS, and the same pass again.
//
//   lane = PSZ * slot + u : row u of patch slot 0 .. NS-1, NS = 64 / PSZ (8 x 8 rows = 64 lanes,
//   5 x 12 = 60, 4 x 16 = 64 ...). One channel at a time,
//   * pass A: a step transforms NS candidates (image patch row in a[], previous-frame patch row in
//     b[] of the same lane), so the statistics of a coefficient are sums over the steps in registers
//     plus one reduction over the slots per channel (through LDS, one statistic at a time; lanes of
//     the first PP/4 slots then own 4 coefficients each and compute their gains). Sums are taken
//     around x0 = the coefficient of the first candidate (as k_group8m.h). A slot without a
//     candidate re-reads candidate 0 and a candidate without a valid previous patch reads
//     candidate 0's image as "previous": bit-identical arithmetic gives exact zeros, so only the
//     transition term needs a mask. The group mean of the previous-frame coefficients is reduced
//     in the step that transforms the last group member;
//   * pass B: a step shrinks and inverts 2 NS group members (filter) or NS (smoother: image and
//     previous patch of a member in one lane); the pixel rows are then staged in LDS slot by slot
//     and added to the private accumulator tile by PSZ rows x NBK blocks of PB pixels of ONE member,
//     so that no two lanes of an instruction touch the same tile entry.
// Coefficients live transposed between the passes (statistics, gains and shrinkage are elementwise).
// A workgroup (one wavefront) owns ONE target. Its accumulator tile holds two planes only: the
// weights and the values of the channel in flight (pass B is channel-major; the plane is flushed to
// HBM with float atomics and cleared after each channel), so that tile + transposition scratch +
// gains stay near 11 KB and the register budget, not LDS, sets the occupancy (the f32 vector ALU
// needs the wavefronts: 4.2 cycles per instruction at 2 per SIMD, 3.0 at 3, profiles/README.md).
//
// The matrix cores are not used here: see k_dct12.h for the arithmetic (12 x 12), and
// profiles/README.md for the measured comparison with k_group8m.h (8 x 8).
#pragma once
#include <type_traits>

#include "k_dct12.h"
#include "k_dct8.h"
#include "k_group8.h"   // NlkGTile, nlk_f4u, nlk_wave_sum8
#include "k_group8m.h"  // nlk_f4, nlk_bperm
#include "nlk_common.h"

// ordering of this wavefront's LDS writes and reads: compiler-only (nlk_common.h); NLK_PP_WAITS
// builds the variant that also waits for the LDS counter (5.22 ms against 5.07 at C3)
#ifdef NLK_PP_WAITS
#define NLK_PP_SYNC nlk_wave_lds_fence
#else
#define NLK_PP_SYNC nlk_wave_lds_order
#endif

template <int PSZ>
struct NlkPP {
  static constexpr int NS = 64 / PSZ;                  // patch slots of a wavefront
  static constexpr int PP = (PSZ + 3) & ~3;            // row pitch in LDS (floats): 16-byte rows
  // floats per slot of the transposition scratch: >= PSZ * PP and = PP (mod 32), so that the
  // columns the slots read sit on disjoint banks
  static constexpr int TS = PSZ * PP + ((PP - PSZ * PP) % 32 + 32) % 32;
  static constexpr int SCRATCH = NS * TS;              // also holds [NS][PSZ][PP] partials / staged rows
  static constexpr int NOWN = PP / 4;                  // slots whose lanes own 4 coefficients each
  static constexpr int PB = (PSZ + NS - 1) / NS;       // aggregation: pixels per lane ...
  static constexpr int NBK = (PSZ + PB - 1) / PB;      // ... and blocks per row (lanes: PSZ x NBK)
  static constexpr int WAVES = PSZ <= 8 ? 4 : (PSZ <= 12 ? 3 : 2);      // wavefronts per SIMD the registers are cut for
  static __host__ __device__ constexpr int gains(int ch) { return ch * 2 * PSZ * PP; }
};

// 1-D transforms of PSZ registers; `basis` = [PSZ][PSZ] orthonormal DCT-II (uniform address:
// scalar loads) for the sizes without a flow graph
template <int N>
__device__ __forceinline__ void nlk_pp_dct_fwd(float (&p)[N], const float* __restrict__ basis) {
  if constexpr (N == 8) nlk_dct8_fast_fwd(p);
  else if constexpr (N == 12) nlk_dct12_fast_fwd(p);
  else if constexpr (N % 2 == 0) {  // C[k][N-1-j] = (-1)^k C[k][j]
    float s[N / 2], d[N / 2], y[N];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { s[i] = p[i] + p[N - 1 - i]; d[i] = p[i] - p[N - 1 - i]; }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float* z = (k & 1) ? d : s;
      float a = basis[k * N] * z[0];
#pragma unroll
      for (int i = 1; i < N / 2; ++i) a = fmaf(basis[k * N + i], z[i], a);
      y[k] = a;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = y[k];
  } else {
    float y[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float a = basis[k * N] * p[0];
#pragma unroll
      for (int j = 1; j < N; ++j) a = fmaf(basis[k * N + j], p[j], a);
      y[k] = a;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = y[k];
  }
}
template <int N>
__device__ __forceinline__ void nlk_pp_dct_inv(float (&y)[N], const float* __restrict__ basis) {
  if constexpr (N == 8) nlk_dct8_fast_inv(y);
  else if constexpr (N == 12) nlk_dct12_fast_inv(y);
  else if constexpr (N % 2 == 0) {
    float E[N / 2], O[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      float e = basis[i] * y[0], o = basis[N + i] * y[1];
#pragma unroll
      for (int k = 2; k < N; k += 2) { e = fmaf(basis[k * N + i], y[k], e); o = fmaf(basis[(k + 1) * N + i], y[k + 1], o); }
      E[i] = e; O[i] = o;
    }
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { y[i] = E[i] + O[i]; y[N - 1 - i] = E[i] - O[i]; }
  } else {
    float x[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float a = basis[j] * y[0];
#pragma unroll
      for (int k = 1; k < N; ++k) a = fmaf(basis[k * N + j], y[k], a);
      x[j] = a;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) y[j] = x[j];
  }
}

// the lane's PSZ values as one row of PP floats in LDS (padding = 0)
template <int PSZ>
__device__ __forceinline__ void nlk_pp_put_row(float* __restrict__ row, const float (&p)[PSZ]) {
  constexpr int PP = NlkPP<PSZ>::PP;
#pragma unroll
  for (int j = 0; j < PP / 4; ++j)
    ((nlk_f4*)row)[j] = nlk_f4{4 * j < PSZ ? p[4 * j < PSZ ? 4 * j : 0] : 0.f,
                               4 * j + 1 < PSZ ? p[4 * j + 1 < PSZ ? 4 * j + 1 : 0] : 0.f,
                               4 * j + 2 < PSZ ? p[4 * j + 2 < PSZ ? 4 * j + 2 : 0] : 0.f,
                               4 * j + 3 < PSZ ? p[4 * j + 3 < PSZ ? 4 * j + 3 : 0] : 0.f};
}

template <int PSZ>
__device__ __forceinline__ void nlk_pp_transpose(float (&p)[PSZ], float* __restrict__ tile, int u, bool on) {
  constexpr int PP = NlkPP<PSZ>::PP;
  if (on) nlk_pp_put_row<PSZ>(tile + PP * u, p);
  NLK_PP_SYNC();
  if (on) {
#pragma unroll
    for (int j = 0; j < PSZ; ++j) p[j] = tile[PP * j + u];
  }
  NLK_PP_SYNC();
}

// PSZ consecutive floats of an image row (exactly PSZ: a patch may end at the image border)
template <int PSZ>
__device__ __forceinline__ void nlk_pp_load_row(const float* __restrict__ p, float (&dst)[PSZ]) {
  typedef const __attribute__((address_space(1))) nlk_f4u* gp4;  // (global, not flat: see k_group8m.h)
  typedef const __attribute__((address_space(1))) float* gp1;
#pragma unroll
  for (int j = 0; j + 3 < PSZ; j += 4) {
    const nlk_f4u v = *(gp4)(p + j);
    dst[j] = v.x; dst[j + 1] = v.y; dst[j + 2] = v.z; dst[j + 3] = v.w;
  }
#pragma unroll
  for (int j = PSZ & ~3; j < PSZ; ++j) dst[j] = *(gp1)(p + j);
}

template <int PSZ, bool SMO>
__global__ void __launch_bounds__(64, NlkPP<PSZ>::WAVES)
k_groupp(const float* __restrict__ img,   // matching / statistics image (planar)
         const float* __restrict__ cur,   // image whose patches are filtered
         const float* __restrict__ prev,  // previous output or nullptr
         NlkGeom g, NlkGTile tl, const uint32_t* __restrict__ topk, const NlkTarget* __restrict__ tinfo,
         const uint32_t* __restrict__ gcoords, const uint8_t* __restrict__ active,
         const float* __restrict__ basis,   // [PSZ][PSZ] orthonormal DCT-II
         const float* __restrict__ window,  // [PSZ][PSZ] aggregation window
         float* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // values | weights | scratch | gains
  typedef NlkPP<PSZ> K;
  constexpr int NS = K::NS, PP = K::PP, TS = K::TS, NOWN = K::NOWN, PB = K::PB, NBK = K::NBK;
  const int step = g.step, CH = g.ch;
  const int lane = threadIdx.x;
  const int ngrid = g.ngx * g.ngy;
  const int ti = nlk_xcd_tile(blockIdx.x, ngrid);  // one target per workgroup
  if (ti >= ngrid) return;
  const NlkTarget info = tinfo[ti];
  const int nagg = info.nagg, k = info.nsel;
  bool work = active[ti] && nagg != 0;
  // (deterministic mode: near and far groups in separate launches)
  if (tl.split && nlk_far_target(g, (info.vbits[0] | info.vbits[1]) ? 1 : 0) != (tl.far != 0)) work = false;
  if (tl.slab && lane == 0) {  // (deterministic mode, k_gather.h)
    tl.tflag[ti] = work;
    if (work) { atomicAdd(&tl.tcount[0], 1); atomicAdd(&tl.tcount[1 + ti / g.ngx], 1); }
  }
  if (!work) return;
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
  float* const gbuf = scratch + K::SCRATCH;  // [CH][gain | (1-gain)*mean][u][PP]

  // lane role: slot (0 .. NS-1; NS = the idle lanes, if any) and row / column u
  const int slot = lane / PSZ, u = lane - PSZ * slot;
  const bool on = slot < NS;
  const int sl = on ? slot : 0;
  float* const tsc = scratch + sl * TS;
#define NLK_PP_FWD(p) do { nlk_pp_dct_fwd<PSZ>(p, basis); nlk_pp_transpose<PSZ>(p, tsc, u, on); nlk_pp_dct_fwd<PSZ>(p, basis); } while (0)
#define NLK_PP_INV(p) do { nlk_pp_dct_inv<PSZ>(p, basis); nlk_pp_transpose<PSZ>(p, tsc, u, on); nlk_pp_dct_inv<PSZ>(p, basis); } while (0)
  // aggregation role (pass B; independent of the transform role): lane = NBK * row + block adds pixels
  // PB*block .. PB*block + PB-1 of row `au` of ONE member. Row-major over the lanes: the 32 lanes of a
  // half-wavefront then read 8 staged rows (pitch PP = 12 floats at 12 x 12: banks 12 au + 3 ablk, all
  // different) - with lane = PSZ * block + row, rows au and au + 8 met on one bank (round 2's 19 %
  // conflict cycles, profiles/README.md round 3)
  const int au = lane / NBK, ablk = lane - NBK * au;
  const bool agg_on = lane < PSZ * NBK;
  float wv[PB];
#pragma unroll
  for (int e = 0; e < PB; ++e) {
    const int x = PB * (agg_on ? ablk : 0) + e;
    wv[e] = (agg_on && x < PSZ) ? window[au * PSZ + x] : 0.f;
  }
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
    it_m = c_last / NS;
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
    const int nit = (k + NS - 1) / NS;
    const int org0 = cand_org(0);
    const bool owner = slot < NOWN;  // lane (slot < NOWN, u) owns coefficients (u, 4*slot .. 4*slot+3)
    float* const red = scratch + (sl * PSZ + u) * PP;                      // partials as [slot][u][PP]
    const float* const rd = scratch + u * PP + 4 * (owner ? slot : 0);
    auto put = [&](const float (&v)[PSZ]) {
      if (on) nlk_pp_put_row<PSZ>(red, v);
      NLK_PP_SYNC();
    };
    auto sum_slots = [&]() -> nlk_f4 {  // sum over the slots of the owned coefficients
      nlk_f4 tsum = nlk_f4{0.f, 0.f, 0.f, 0.f};
      if (owner) {
#pragma unroll
        for (int s5 = 0; s5 < NS; ++s5) tsum += *(const nlk_f4*)(rd + s5 * PSZ * PP);
      }
      NLK_PP_SYNC();
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
        // (the filter's Kalman branch uses no image statistics, reference :859-904: three sums there)
        constexpr bool IMG = !HP || SMO;
        constexpr int NSUM = (IMG ? 2 : 0) + (HP ? 3 : 0), P0 = IMG ? 2 : 0;  // previous-frame sums from P0
        float S[NSUM][PSZ], x0[PSZ];
#pragma unroll
        for (int a = 0; a < NSUM; ++a)
#pragma unroll
          for (int r = 0; r < PSZ; ++r) S[a][r] = 0.f;
        nlk_f4 tot5 = nlk_f4{0.f, 0.f, 0.f, 0.f};  // previous frame over the group members (owners)
        // the slot's candidate in step `it`: validity mask and the addresses of its two rows
        auto job = [&](int it, float& vm, const float*& pa, const float*& pb) {
          const int ci = NS * it + sl;
          const bool valid = on && ci < k;
          const int cl = valid ? ci : 0;
          const uint64_t vw = cl < 64 ? vbits[0] : vbits[1];
          const bool v = valid && ((vw >> (cl & 63)) & 1ull);
          vm = v ? 1.f : 0.f;
          const int org = cand_org(cl);
          // no valid previous patch: the "previous" rows are candidate 0's image rows (= x0: exact zeros in the
          // sums), and in the filter - whose Kalman branch uses no image statistics - the image rows too, so
          // that the transition term is an exact zero as well and needs no mask
          pa = (HP && !SMO && !v) ? img_c + org0 : img_c + org;
          pb = v ? img_c + prev_off + org : img_c + org0;
        };
        // one step: the rows in (a, b) are transformed and accumulated while the next step's rows
        // travel into (na, nb); the two register sets swap roles from step to step
        auto stage = [&](int it, float vm, float (&a)[PSZ], float (&b)[PSZ], float& nvm, float (&na)[PSZ],
                         float (&nb)[PSZ]) {
          const float *pa, *pb;
          job(it + 1 < nit ? it + 1 : it, nvm, pa, pb);
          nlk_pp_load_row<PSZ>(pa, na);
          if (HP) nlk_pp_load_row<PSZ>(pb, nb);
          NLK_PP_FWD(a);
          if (HP) NLK_PP_FWD(b);
          if (it == 0) {
#pragma unroll
            for (int r = 0; r < PSZ; ++r) x0[r] = nlk_bperm(a[r], u);  // slot 0 holds candidate 0
          }
          if (HP && !SMO && it == it_m) {
            // group mean: previous-frame deviations of the members so far + this step's members
            const int ci = NS * it + sl;
            const uint64_t vw = ci < 64 ? vbits[0] : vbits[1];
            const int rank = (ci < 64 ? 0 : np0a) + __popcll(vw & ((1ull << (ci & 63)) - 1ull));
            const float gm = (vm != 0.f && rank < g.ntagg) ? 1.f : 0.f;
            if (on) {  // (four values at a time: PSZ more live registers would not fit)
#pragma unroll
              for (int j = 0; j < PP / 4; ++j) {
                nlk_f4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int r = 4 * j + e < PSZ ? 4 * j + e : 0;
                  v4[e] = 4 * j + e < PSZ ? fmaf(gm, b[r] - x0[r], S[HP ? P0 : 0][r]) : 0.f;
                }
                ((nlk_f4*)red)[j] = v4;
              }
            }
            NLK_PP_SYNC();
            tot5 = sum_slots();
          }
          if constexpr (HP) {
#pragma unroll
            for (int r = 0; r < PSZ; ++r) {
              const float db = b[r] - x0[r];
              if constexpr (IMG) {
                const float da = a[r] - x0[r];
                S[0][r] += da;
                S[1][r] = fmaf(da, da, S[1][r]);
              }
              S[P0][r] += db;
              S[P0 + 1][r] = fmaf(db, db, S[P0 + 1][r]);
              const float df = b[r] - a[r];  // reference: :769-783, smoother :1659-1667
              if constexpr (SMO) S[P0 + 2][r] = fmaf(vm * df, df, S[P0 + 2][r]);
              else S[P0 + 2][r] = fmaf(df, df, S[P0 + 2][r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < PSZ; ++r) {
              const float da = a[r] - x0[r];
              S[0][r] += da;
              S[1][r] = fmaf(da, da, S[1][r]);
            }
          }
        };
        float A1[PSZ], B1[PSZ], A2[PSZ], B2[PSZ], vm1, vm2 = 0.f;
#pragma unroll
        for (int r = 0; r < PSZ; ++r) B1[r] = B2[r] = 0.f;
        {
          const float *pa, *pb;
          job(0, vm1, pa, pb);
          nlk_pp_load_row<PSZ>(pa, A1);
          if (HP) nlk_pp_load_row<PSZ>(pb, B1);
        }
        for (int it = 0; it < nit; it += 2) {
          stage(it, vm1, A1, B1, vm2, A2, B2);
          if (it + 1 < nit) stage(it + 1, vm2, A2, B2, vm1, A1, B1);
        }
        // ---- sums over the slots, one statistic at a time
        nlk_f4 tot[5], x04;
        put(x0);
        x04 = owner ? *(const nlk_f4*)rd : nlk_f4{0.f, 0.f, 0.f, 0.f};  // (every slot holds the same x0)
        NLK_PP_SYNC();
#pragma unroll
        for (int st = 0; st < 5; ++st) {  // tot[0..1] image, tot[2..4] previous frame
          tot[st] = nlk_f4{0.f, 0.f, 0.f, 0.f};
          const int si = st < 2 ? (IMG ? st : -1) : (HP ? P0 + st - 2 : -1);
          if (si >= 0) {
            put(S[si >= 0 && si < NSUM ? si : 0]);
            tot[st] = sum_slots();
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
            if (PP == PSZ || 4 * slot + j < PSZ) part_sum += term;  // (padding coefficients own nothing)
            ga4[j] = ga;
            mu4[j] = (1 - ga) * m;  // filter: a*PG + (1-a)*M (reference: :879, :902)
          }
          *(nlk_f4*)(gbuf + ((c * 2 + 0) * PSZ + u) * PP + 4 * slot) = ga4;
          *(nlk_f4*)(gbuf + ((c * 2 + 1) * PSZ + u) * PP + 4 * slot) = mu4;
        }
        NLK_PP_SYNC();
      }
    };
    if (hp) pass_a(std::true_type{});
    else pass_a(std::false_type{});
  }
  // the reference adds the same per-coefficient terms once per group member
  float vp = nlk_wave_sum8(part_sum) * (float)nagg;
  if (passthrough) vp = 0.f;
  const float wgt = 1.f / (vp > 1e-6f ? vp : 1e-6f);
  float ww[PB];
#pragma unroll
  for (int e = 0; e < PB; ++e) ww[e] = wgt * wv[e];
  // where every member lands in the tile, once per target with one member per lane (as k_group8m.h): its
  // offset (floats) inside a plane and an "inside the tile" bit (entries past the last member count as inside)
  uint32_t mbase[2];
  uint64_t inside[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int lx = nlk_x(greg[m]) - rx0, ly = nlk_y(greg[m]) - ry0;
    const bool in = lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh;
    mbase[m] = (uint32_t)(ly * rwp + lx);
    inside[m] = __ballot(in || lane + 64 * m >= nagg);
  }

  // ---------------- pass B: shrink, invert and aggregate the group members
  // rows staged in `scratch` as [slot][u][PP]; member n0 + s of the round is added by PSZ x NBK lanes
  auto add_round = [&](float (&px)[PSZ], int n0, int c) {
    if (on) nlk_pp_put_row<PSZ>(scratch + (sl * PSZ + u) * PP, px);
    NLK_PP_SYNC();
    // a round whose NS members are all inside the tile (every round of a temporal target: the tile's halo is the
    // temporal radius) is added in straight-line code: no per-member in-tile test, list select or loop branch
    {
      const int b0 = n0 & 63;
      const uint64_t im = n0 < 64 ? inside[0] : inside[1];
      if (b0 + NS <= 64 && n0 + NS <= nagg && ((im >> b0) & ((1ull << NS) - 1ull)) == (1ull << NS) - 1ull) {
        const uint32_t mb = n0 < 64 ? mbase[0] : mbase[1];
        int toff[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) toff[s] = __builtin_amdgcn_readlane((int)mb, b0 + s);  // (every lane active here)
        if (agg_on) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const float* sp = scratch + (s * PSZ + au) * PP + PB * ablk;
            float* dst = vplane + toff[s] + au * rwp + PB * ablk;
            float v[PB], o[PB];
#pragma unroll
            for (int e = 0; e < PB; ++e) v[e] = sp[e];
#pragma unroll
            for (int e = 0; e < PB; ++e) o[e] = PB * ablk + e < PSZ ? dst[e] : 0.f;
#pragma unroll
            for (int e = 0; e < PB; ++e)
              if (PB * ablk + e < PSZ) dst[e] = fmaf(ww[e], v[e], o[e]);
            if (c == 0) {
              float* dw = wplane + toff[s] + au * rwp + PB * ablk;
#pragma unroll
              for (int e = 0; e < PB; ++e) o[e] = PB * ablk + e < PSZ ? dw[e] : 0.f;
#pragma unroll
              for (int e = 0; e < PB; ++e)
                if (PB * ablk + e < PSZ) dw[e] = o[e] + ww[e];
            }
          }
        }
        NLK_PP_SYNC();
        return;
      }
    }
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      const int mi = n0 + s;
      if (mi >= nagg) break;
      const uint32_t q = mi < 64 ? __builtin_amdgcn_readlane(greg[0], mi)
                                 : __builtin_amdgcn_readlane(greg[1], mi - 64);
      const int qx = nlk_x(q), qy = nlk_y(q);
      const int lx = qx - rx0, ly = qy - ry0;
      if (agg_on) {
        const float* sp = scratch + (s * PSZ + au) * PP + PB * ablk;
        float v[PB];
#pragma unroll
        for (int e = 0; e < PB; ++e) v[e] = sp[e];  // (reads of the row's padding where PB*slot + e >= PSZ: weight 0)
        if (lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh) {
          float* dst = vplane + (ly + au) * rwp + lx + PB * ablk;
          float o[PB];
#pragma unroll
          for (int e = 0; e < PB; ++e) o[e] = PB * ablk + e < PSZ ? dst[e] : 0.f;
#pragma unroll
          for (int e = 0; e < PB; ++e)
            if (PB * ablk + e < PSZ) dst[e] = fmaf(ww[e], v[e], o[e]);
          if (c == 0) {
            float* dw = wplane + (ly + au) * rwp + lx + PB * ablk;
#pragma unroll
            for (int e = 0; e < PB; ++e) o[e] = PB * ablk + e < PSZ ? dw[e] : 0.f;
#pragma unroll
            for (int e = 0; e < PB; ++e)
              if (PB * ablk + e < PSZ) dw[e] = o[e] + ww[e];
          }
        } else {
          float* dst = acc + (size_t)c * npix + (size_t)(qy + au) * g.w + qx + PB * ablk;
#pragma unroll
          for (int e = 0; e < PB; ++e)
            if (PB * ablk + e < PSZ) unsafeAtomicAdd(dst + e, ww[e] * v[e]);
          if (c == 0) {
            float* dw = acc + (size_t)CH * npix + (size_t)(qy + au) * g.w + qx + PB * ablk;
#pragma unroll
            for (int e = 0; e < PB; ++e)
              if (PB * ablk + e < PSZ) unsafeAtomicAdd(dw + e, ww[e]);
          }
        }
      }
    }
    NLK_PP_SYNC();
  };
  // a tile plane -> HBM (coalesced rows, untouched entries skipped), cleared for the next channel
  // (a narrow tile puts two or four rows on the 64 lanes: fewer, fuller atomic instructions)
  // (the region walked as one run of rw x rh entries, 64 per atomic instruction: a wavefront's flush is paced by
  // the atomics it may have in flight, i.e. by their number - k_group8m.h)
  const int qstep = 64 / rw, rstep = 64 - qstep * rw;
  const int fy0 = lane / rw, fx0 = lane - fy0 * rw;
  auto flush = [&](float* sp, int p, bool clear) {
    if (tl.slab) {  // deterministic mode: the plane as it stands, into this target's slab
      nlk_f4* dst = reinterpret_cast<nlk_f4*>(tl.slab + ((size_t)ti * (CH + 1) + p) * plane);
      for (int i = lane; i < plane / 4; i += 64) {
        dst[i] = reinterpret_cast<const nlk_f4*>(sp)[i];
        if (clear) reinterpret_cast<nlk_f4*>(sp)[i] = nlk_f4{0.f, 0.f, 0.f, 0.f};
      }
      NLK_PP_SYNC();
      return;
    }
    float* dp = acc + (size_t)p * npix + (size_t)ry0 * g.w + rx0;
    int y = fy0, xx = fx0;
#pragma unroll 2
    for (; y < rh; ) {
      const float v = sp[y * rwp + xx];
      if (v != 0.f) {
        unsafeAtomicAdd(dp + (size_t)y * g.w + xx, v);
        if (clear) sp[y * rwp + xx] = 0.f;
      }
      xx += rstep; y += qstep;
      if (xx >= rw) { xx -= rw; ++y; }
    }
    NLK_PP_SYNC();
  };
  // The member rows of a step are requested one step ahead (also across the change of channel), and
  // a channel's plane is flushed only after the next step's rows have ARRIVED: vector-memory
  // operations complete in issue order, so rows requested behind some sixty flush atomics would wait
  // for every one of them (microseconds under load); this way nothing ever waits for an atomic.
  const int per = SMO ? NS : 2 * NS;                          // members per step
  const int nst = passthrough ? 1 : (nagg + per - 1) / per;   // steps per channel
  float a[PSZ], b[PSZ], na[PSZ], nb[PSZ], gain[PSZ], mu[PSZ];
#pragma unroll
  for (int r = 0; r < PSZ; ++r) { b[r] = nb[r] = 0.f; gain[r] = mu[r] = 0.f; }
  auto request = [&](int c, int st, float (&ra)[PSZ], float (&rb)[PSZ]) {  // rows of step `st` of channel c
    const float* img_c = img + c * npix;
    const int n0 = st * per;
    if (passthrough) {
      nlk_pp_load_row<PSZ>(img_c + src_off + memb_org(0), ra);
    } else if (!SMO) {  // member n0 + slot in a[], member n0 + NS + slot in b[]
      const int ma = min(n0 + sl, nagg - 1), mb = min(n0 + NS + sl, nagg - 1);
      nlk_pp_load_row<PSZ>(img_c + src_off + memb_org(ma), ra);
      nlk_pp_load_row<PSZ>(img_c + src_off + memb_org(mb), rb);
    } else {            // image and previous-frame patch of member n0 + slot
      const int org = memb_org(min(n0 + sl, nagg - 1));
      nlk_pp_load_row<PSZ>(img_c + src_off + org, ra);
      nlk_pp_load_row<PSZ>(img_c + prev_off + org, rb);
    }
  };
  request(0, 0, a, b);
  for (int c = 0; c < CH; ++c) {
    if (!passthrough) {
      const float* gp = gbuf + ((c * 2 + 0) * PSZ + u) * PP;
      const float* mp = gbuf + ((c * 2 + 1) * PSZ + u) * PP;
#pragma unroll
      for (int j = 0; j < PP / 4; ++j) {
        const nlk_f4 gv = ((const nlk_f4*)gp)[j], mv = ((const nlk_f4*)mp)[j];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * j + e < PSZ) { gain[4 * j + e < PSZ ? 4 * j + e : 0] = gv[e]; mu[4 * j + e < PSZ ? 4 * j + e : 0] = mv[e]; }
      }
    }
    for (int st = 0; st < nst; ++st) {
      const int n0 = st * per;
      const bool last_of_channel = st + 1 == nst, more = !last_of_channel || c + 1 < CH;
      if (passthrough) {
      } else if (!SMO) {
        const bool two = n0 + NS < nagg;
        NLK_PP_FWD(a);
        if (two) NLK_PP_FWD(b);
#pragma unroll
        for (int r = 0; r < PSZ; ++r) {
          a[r] = fmaf(gain[r], a[r], mu[r]);
          b[r] = fmaf(gain[r], b[r], mu[r]);
        }
        NLK_PP_INV(a);
        if (two) NLK_PP_INV(b);
      } else {
        NLK_PP_FWD(a);
        NLK_PP_FWD(b);
#pragma unroll
        for (int r = 0; r < PSZ; ++r) a[r] = (1 - gain[r]) * a[r] + gain[r] * b[r];  // reference: :1775
        NLK_PP_INV(a);
      }
      if (more) request(last_of_channel ? c + 1 : c, last_of_channel ? 0 : st + 1, na, nb);
      add_round(a, n0, c);
      if (!passthrough && !SMO && n0 + NS < nagg) add_round(b, n0 + NS, c);
      if (last_of_channel) {
        // (the requested rows are consumed here, i.e. waited for, BEFORE the flush's atomics are issued)
#pragma unroll
        for (int r = 0; r < PSZ; ++r) { asm volatile("" : "+v"(na[r])); asm volatile("" : "+v"(nb[r])); }
        flush(vplane, c, c + 1 < CH);
      }
#pragma unroll
      for (int r = 0; r < PSZ; ++r) { a[r] = na[r]; b[r] = nb[r]; }
    }
  }
  flush(wplane, CH, false);
}

#undef NLK_PP_FWD
#undef NLK_PP_INV
