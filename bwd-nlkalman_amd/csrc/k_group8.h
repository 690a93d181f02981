// k_group8.h — group processing for 8x8 patches, the register/DPP fast path
// (reference: src/nlkalman.c:713-932 filter, :1603-1845 smoother).
//
// Layout. A wavefront processes one target at a time. Lane l = 16*c + 8*sel + u
// holds ROW u of one 8x8 plane in 8 registers: channel c, sel = 0 the matching
// image / the patch being filtered, sel = 1 the previous frame. (Lanes with
// c >= CH idle.) With this layout
//   * the row pass of the separable DCT is pure register arithmetic
//     (even/odd split, 8 adds + 32 FMAs for 8 outputs),
//   * the column pass crosses only the 8 lanes of a plane: lane u needs
//     T[u ^ k], k = 0..7, which are the quad_perm / row_half_mirror DPP
//     patterns, so it is 1 v_mov_dpp + 1 v_mul + 7 v_fmac per register, no LDS,
//   * every DCT coefficient stays in the same (lane, register) for all
//     candidates, so the Welford statistics, gains and shrinkage are
//     lane-local; the only other cross-lane traffic is the partner exchange
//     img <-> prev (lane ^ 8, DPP row_ror:8) for the transition variance.
// Group members are re-transformed once the gains are known (two members per
// step in the filter: one in the sel = 0 lanes, one in the sel = 1 lanes).
//
// Aggregation. A workgroup is ONE wavefront that owns a tile of 4 x 1 targets
// and a private LDS accumulator tile (ch value planes + 1 weight plane, covering
// the tile plus the search halo). Being private, the tile is updated with plain
// ds_read / add / ds_write (LDS float atomics retire about one lane per clock on
// gfx950 and made an earlier, shared-tile version of this kernel LDS-bound: see
// profiles/). The tile is flushed once with coalesced global float atomics,
// skipping untouched entries, which cuts the HBM atomic traffic ~10x compared
// with one global atomic per patch pixel (global float atomics run at a fixed
// ~1.3 TB/s of added bytes on MI355X and would otherwise bound the kernel).
#pragma once
#include "nlk_common.h"

constexpr float NLK_C8[8][8] = {
    {0.353553385f, 0.353553385f, 0.353553385f, 0.353553385f, 0.353553385f, 0.353553385f, 0.353553385f, 0.353553385f},
    {0.490392625f, 0.415734798f, 0.277785122f, 0.0975451618f, -0.0975451618f, -0.277785122f, -0.415734798f, -0.490392625f},
    {0.461939752f, 0.191341713f, -0.191341713f, -0.461939752f, -0.461939752f, -0.191341713f, 0.191341713f, 0.461939752f},
    {0.415734798f, -0.0975451618f, -0.490392625f, -0.277785122f, 0.277785122f, 0.490392625f, 0.0975451618f, -0.415734798f},
    {0.353553385f, -0.353553385f, -0.353553385f, 0.353553385f, 0.353553385f, -0.353553385f, -0.353553385f, 0.353553385f},
    {0.277785122f, -0.490392625f, 0.0975451618f, 0.415734798f, -0.415734798f, -0.0975451618f, 0.490392625f, -0.277785122f},
    {0.191341713f, -0.461939752f, 0.461939752f, -0.191341713f, -0.191341713f, 0.461939752f, -0.461939752f, 0.191341713f},
    {0.0975451618f, -0.277785122f, 0.415734798f, -0.490392625f, 0.490392625f, -0.415734798f, 0.277785122f, -0.0975451618f},
};

struct NlkGTile {
  int tgx, tgy, ntx, nty;
  // k_group8m: tile rows of tgy grid rows; the tile rows after them hold ONE grid row each, and the last `single`
  // grid rows are worked through one target per workgroup, in the workgroups after `nmain` (end of the launch)
  int nty_full, single, nmain;
  int rwp, rh_max;  // LDS accumulator region: row stride / rows
  int wmax;         // halo of the LDS tile around its targets (groups reaching further spill to HBM atomics)
  int plane;        // k_group8m / k_groupp: floats per accumulator plane (padded, see there)
  // deterministic aggregation (k_gather.h): where workgroup `tile` writes its planes instead of adding
  // them to the frame accumulator with atomics, and the flag that says it did
  float* slab;      // [tiles][planes][plane]
  uint8_t* tflag;   // [tiles]
  int* tcount;      // [1 + nty]: flagged tiles of the launch / of every tile row (zeroed by the host)
  // A temporal frame has two kinds of groups: those searched with the temporal radius and the few
  // spatial-branch ones (no valid previous patch) that reach wsz_x. Deterministic mode runs them in
  // two launches with a tile halo each (far = 0: the former only, far = 1: the latter only), so that
  // no member ever leaves its tile.
  int split, far;   // split != 0: this launch takes only the targets of kind `far`
  // k_group8m: start of the allocation that holds every planar image of the call (nlk_ctx::planes, < 4 GiB):
  // patches are addressed as this base + a 32-bit byte offset
  const float* pbase;
  const float* diff;   // k_group8m, smoother: planar (previous - image) inside the same slab, or nullptr (then the kernel subtracts)
  // k_group8m, mask replay inside the launch (chase != 0): workgroup 0 first replays the processed mask of the grid
  // rows [0, chase_rows) from the bit planes (k_commit_rows.h) and publishes every row's decisions as
  // generation-tagged words; every workgroup polls the words of its own targets instead of reading `active`.
  // The launch's own first grid row is row chase_row0 of that grid (0 for a whole-frame call, a strip's first row).
  int chase;                     // 0, or the reach (1..3) of the grid whose replay this launch runs
  int chase_test_skip0;          // test hook (NLK_CHASE_TEST_SKIP0=1): workgroup 0 does NOT replay - every workgroup
                                 // must get its decisions by replaying the rows it needs itself
  int chase_row0, chase_rows;
  const uint32_t* chase_gen;     // the generation of this launch: a device word the bit-plane kernel has just advanced
                                 // (a kernel argument would be frozen into a captured HIP graph)
  const uint32_t* chase_planes;  // [rows][planes of that reach][64]
  uint64_t* chase_words;         // [rows][64]: generation << 32 | decision bits
};

// does this target's group reach beyond the temporal radius? (np0 = its candidates with a valid previous patch)
__device__ __forceinline__ bool nlk_far_target(const NlkGeom& g, int np0) {
  return g.have_prev && !g.smoother && np0 == 0;
}

typedef float nlk_f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int CTRL>
__device__ __forceinline__ float nlk_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
#define NLK_DPP_XOR1 0xB1   // quad_perm [1,0,3,2]
#define NLK_DPP_XOR2 0x4E   // quad_perm [2,3,0,1]
#define NLK_DPP_XOR3 0x1B   // quad_perm [3,2,1,0]
#define NLK_DPP_HMIRROR 0x141  // lane u <- lane u ^ 7 inside each group of 8
#define NLK_DPP_ROR8 0x128     // lane l <- lane l ^ 8 inside each row of 16

// forward 1-D DCT-II of 8 registers (even/odd split)
__device__ __forceinline__ void nlk_dct8_fwd(float (&p)[8]) {
  float s[4], d[4], y[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { s[i] = p[i] + p[7 - i]; d[i] = p[i] - p[7 - i]; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float* z = (k & 1) ? d : s;
    float a = NLK_C8[k][0] * z[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) a = fmaf(NLK_C8[k][i], z[i], a);
    y[k] = a;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = y[k];
}

// inverse (DCT-III): x[i] = sum_k C[k][i] y[k]
__device__ __forceinline__ void nlk_dct8_inv(float (&y)[8]) {
  float E[4], O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float e = NLK_C8[0][i] * y[0], o = NLK_C8[1][i] * y[1];
#pragma unroll
    for (int k = 2; k < 8; k += 2) { e = fmaf(NLK_C8[k][i], y[k], e); o = fmaf(NLK_C8[k + 1][i], y[k + 1], o); }
    E[i] = e; O[i] = o;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { y[i] = E[i] + O[i]; y[7 - i] = E[i] - O[i]; }
}

// Column pass across the 8 lanes of a plane, four registers at a time:
//   p_i[u] <- sum_k ck[k] * p_i[u ^ k]
// hipcc (ROCm 7.2) does not fold update_dpp into the consuming FMA on gfx950, so
// the DPP forms are written out: per register 1 v_mov_dpp (u ^ 7), 1 v_mul and
// 7 v_fmac (6 of them DPP). The s_nop at both ends covers the "VALU write ->
// DPP read, 2 wait states" hazard against the surrounding compiler code, which
// does not see inside an asm statement.
#define NLK_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define NLK_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#define NLK_QP3 "quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
#define NLK_HM "row_half_mirror row_mask:0xf bank_mask:0xf"

__device__ __forceinline__ void nlk_col8x4(float& p0, float& p1, float& p2, float& p3,
                                           const float (&ck)[8]) {
  float y0, y1, y2, y3, m0, m1, m2, m3;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32_dpp %4, %8 " NLK_HM "\n\t"
      "v_mov_b32_dpp %5, %9 " NLK_HM "\n\t"
      "v_mov_b32_dpp %6, %10 " NLK_HM "\n\t"
      "v_mov_b32_dpp %7, %11 " NLK_HM "\n\t"
      "v_mul_f32 %0, %12, %8\n\t"
      "v_mul_f32 %1, %12, %9\n\t"
      "v_mul_f32 %2, %12, %10\n\t"
      "v_mul_f32 %3, %12, %11\n\t"
      "v_fmac_f32_dpp %0, %8, %13 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %1, %9, %13 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %2, %10, %13 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %3, %11, %13 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %0, %8, %14 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %1, %9, %14 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %2, %10, %14 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %3, %11, %14 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %0, %8, %15 " NLK_QP3 "\n\t"
      "v_fmac_f32_dpp %1, %9, %15 " NLK_QP3 "\n\t"
      "v_fmac_f32_dpp %2, %10, %15 " NLK_QP3 "\n\t"
      "v_fmac_f32_dpp %3, %11, %15 " NLK_QP3 "\n\t"
      "v_fmac_f32 %0, %4, %19\n\t"
      "v_fmac_f32 %1, %5, %19\n\t"
      "v_fmac_f32 %2, %6, %19\n\t"
      "v_fmac_f32 %3, %7, %19\n\t"
      "v_fmac_f32_dpp %0, %4, %18 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %1, %5, %18 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %2, %6, %18 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %3, %7, %18 " NLK_QP1 "\n\t"
      "v_fmac_f32_dpp %0, %4, %17 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %1, %5, %17 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %2, %6, %17 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %3, %7, %17 " NLK_QP2 "\n\t"
      "v_fmac_f32_dpp %0, %4, %16 " NLK_QP3 "\n\t"
      "v_fmac_f32_dpp %1, %5, %16 " NLK_QP3 "\n\t"
      "v_fmac_f32_dpp %2, %6, %16 " NLK_QP3 "\n\t"
      "v_fmac_f32_dpp %3, %7, %16 " NLK_QP3 "\n\t"
      "s_nop 1"
      : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(ck[0]), "v"(ck[1]), "v"(ck[2]), "v"(ck[3]),
        "v"(ck[4]), "v"(ck[5]), "v"(ck[6]), "v"(ck[7]));
  p0 = y0; p1 = y1; p2 = y2; p3 = y3;
}

__device__ __forceinline__ void nlk_dct8x8_fwd(float (&p)[8], const float (&ck)[8]) {
  nlk_dct8_fwd(p);
  nlk_col8x4(p[0], p[1], p[2], p[3], ck);
  nlk_col8x4(p[4], p[5], p[6], p[7], ck);
}
__device__ __forceinline__ void nlk_dct8x8_inv(float (&p)[8], const float (&cik)[8]) {
  nlk_col8x4(p[0], p[1], p[2], p[3], cik);
  nlk_col8x4(p[4], p[5], p[6], p[7], cik);
  nlk_dct8_inv(p);
}

__device__ __forceinline__ void nlk_load_row8(const float* __restrict__ p, float (&dst)[8]) {
  const nlk_f4u a = *reinterpret_cast<const nlk_f4u*>(p);
  const nlk_f4u b = *reinterpret_cast<const nlk_f4u*>(p + 4);
  dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
  dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
}

__device__ __forceinline__ float nlk_wave_sum8(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

#ifndef NLK_G8_WAVES
#define NLK_G8_WAVES 4
#endif
template <int CH, bool SMO>
__global__ void __launch_bounds__(64, NLK_G8_WAVES)
k_group8(const float* __restrict__ img,   // matching / statistics image (planar)
         const float* __restrict__ cur,   // image whose patches are filtered
         const float* __restrict__ prev,  // previous output or nullptr
         const uint8_t* __restrict__ vmap, NlkGeom g, NlkGTile tl,
         const uint32_t* __restrict__ topk, const NlkTarget* __restrict__ tinfo,
         const uint32_t* __restrict__ gcoords, const uint8_t* __restrict__ active,
         const float* __restrict__ basis,   // [8][8] orthonormal DCT-II
         const float* __restrict__ window,  // [8][8] aggregation window
         float* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [(CH+1)][rh_max][rwp]
  constexpr int PSZ = 8, step = 4;
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
  const int u = lane & 7, sel = (lane >> 3) & 1, c = lane >> 4;
  const bool lane_on = c < CH;
  const int cc = lane_on ? c : 0;
  float ck[8], cik[8], wrow[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ck[k] = basis[u * 8 + (u ^ k)];
    cik[k] = basis[(u ^ k) * 8 + u];
    wrow[k] = window[u * 8 + k];
  }
  const size_t npix = (size_t)g.w * g.h;
  const float* img_c = img + cc * npix + (size_t)u * g.w;
  const float* cur_c = cur + cc * npix + (size_t)u * g.w;
  const float* prev_c = prev ? prev + cc * npix + (size_t)u * g.w : nullptr;
  const float* src_c = g.have_basic ? cur_c : img_c;  // patches that get filtered
  const float s2 = g.sigma2;

  // per-tile records: lane tt holds the record of target tt, read back with
  // v_readlane so that no HBM latency sits in the target loop
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
    const int gx = gx0 + tx, gy = gy0 + ty;
    const size_t t = (size_t)gy * g.ngx + gx;
    const bool prev_p = __builtin_amdgcn_readlane(rec_flags, tt) & 1;
    const int k = __builtin_amdgcn_readlane(rec_nsel, tt);

    // candidate / member lists of this target: one entry per lane (two rounds),
    // fetched once; the loops below read them with v_readlane
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
    float mean[8], var[8], v01[8], m0[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) mean[r] = var[r] = v01[r] = m0[r] = 0.f;
    int np0 = 0, np1 = 0;
    float x[8], nxt[8];
    bool vnext = false;
    auto gatherA = [&](int i, float (&dst)[8], bool& v) {
      const uint32_t q = cand(i);
      const int org = nlk_y(q) * g.w + nlk_x(q);
      v = (vbits[i >> 6] >> (i & 63)) & 1ull;
      if (lane_on && (sel == 0 || v)) nlk_load_row8((sel ? prev_c : img_c) + org, dst);
      else {
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[r] = 0.f;
      }
    };
    if (k > 0) gatherA(0, nxt, vnext);
    for (int i = 0; i < k; ++i) {
      const bool v = vnext;
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = nxt[r];
      if (i + 1 < k) gatherA(i + 1, nxt, vnext);
      nlk_dct8x8_fwd(x, ck);
      np1++;
      if (v) np0++;
      const float inv = sel ? (v ? __builtin_amdgcn_rcpf((float)np0) : 0.f) : __builtin_amdgcn_rcpf((float)np1);
      const bool in_group = v && np0 <= g.ntagg;
      const bool upd = lane_on && (sel == 0 || v);
      // difference to the partner plane's coefficient (all lanes take part in the DPP);
      // one v_sub_f32_dpp per register (hipcc would emit v_mov_dpp + v_sub)
      float dif[8];
      asm volatile(
          "s_nop 1\n\t"
          "v_sub_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "v_sub_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1"
          : "=&v"(dif[0]), "=&v"(dif[1]), "=&v"(dif[2]), "=&v"(dif[3]), "=&v"(dif[4]),
            "=&v"(dif[5]), "=&v"(dif[6]), "=&v"(dif[7])
          : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
      if (upd) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float d = x[r] - mean[r];
          mean[r] = fmaf(d, inv, mean[r]);
          var[r] = fmaf(d, x[r] - mean[r], var[r]);
        }
        if (sel) {  // previous-frame lanes (reference: :769-783, smoother :1659-1667)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            v01[r] = fmaf(dif[r], dif[r], v01[r]);
          }
          if (!SMO && in_group) {
#pragma unroll
            for (int r = 0; r < 8; ++r) m0[r] = fmaf(x[r] - m0[r], inv, m0[r]);
          }
        }
      }
    }

    // ---------------- gains (reference: :799-811, :859-904; smoother :1683-1776)
    // owner lanes hold the statistics the gain is made of: the previous-frame
    // lanes when np0 > 0 (Kalman / smoother), the image lanes otherwise (Wiener)
    const bool own = lane_on && (sel == (np0 > 0 ? 1 : 0));
    float gain[8], mu[8];
    float part_sum = 0.f;
    {
      const float in1 = np1 ? 1.f / (float)np1 : 0.f;
      const float in0 = np0 ? 1.f / (float)np0 : 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float vn = var[r] * (sel ? in0 : in1);    // V0 (sel=1) or V1 (sel=0)
        const float v01n = v01[r] * in0;
        const float v1p = nlk_dpp<NLK_DPP_ROR8>(vn);    // V1 seen from the previous-frame lanes
        float a, term, m;
        if (SMO) {
          a = v1p / (v1p + g.beta_t * v01n);
          const float pv = vn - g.beta_t * v01n;
          term = (1 - a * a) * v1p + a * a * (pv > 0.f ? pv : 0.f);
          m = 0.f;
        } else if (np0 > 0) {
          const float d = v01n - (g.have_basic ? 0.f : s2);
          const float v = vn + (0.f > d ? 0.f : d);
          a = v / (v + g.beta_t * s2);
          term = (1 - a * a) * v + a * a * s2;
          m = m0[r];
        } else {
          const float d = vn - (g.have_basic ? 0.f : s2);
          const float v = 0.f > d ? 0.f : d;
          a = v / (v + g.beta_x * s2);
          term = a * v;
          m = mean[r];
        }
        if (own) part_sum += term;
        // make gain and mean available in both lane groups of the channel
        const float a_o = nlk_dpp<NLK_DPP_ROR8>(a), m_o = nlk_dpp<NLK_DPP_ROR8>(m);
        gain[r] = own ? a : a_o;
        mu[r] = (1 - gain[r]) * (own ? m : m_o);  // filter: a*PG + (1-a)*M (reference: :879, :902)
      }
    }
    // the reference adds the same per-coefficient terms once per group member
    float vp = nlk_wave_sum8(part_sum) * (float)nagg;
    const bool passthrough = SMO && np0 == 0;  // reference: :1795-1804
    if (passthrough) vp = 0.f;
    const float wgt = 1.f / (vp > 1e-6f ? vp : 1e-6f);

    // ---------------- pass B: shrink, invert and aggregate the group members
    // A member inside the LDS tile (the rule) is added with LDS atomics; a
    // member of a far-reaching group that leaves it goes straight to HBM.
    auto add_patch = [&](int qx, int qy, const float (&px8)[8]) {
      const int lx = qx - rx0, ly = qy - ry0;
      if (lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh) {
        float* dst = smem + cc * plane + (ly + u) * rwp + lx;
        float* dw = smem + CH * plane + (ly + u) * rwp + lx;
        float o[8], ow[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { o[r] = dst[r]; if (c == 0) ow[r] = dw[r]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float ww = wgt * wrow[r];
          dst[r] = o[r] + ww * px8[r];
          if (c == 0) dw[r] = ow[r] + ww;
        }
      } else {
        float* dst = acc + (size_t)cc * npix + (size_t)(qy + u) * g.w + qx;
        float* dw = acc + (size_t)CH * npix + (size_t)(qy + u) * g.w + qx;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float ww = wgt * wrow[r];
          unsafeAtomicAdd(dst + r, ww * px8[r]);
          if (c == 0) unsafeAtomicAdd(dw + r, ww);
        }
      }
    };
    if (!SMO) {
      for (int n0 = 0; n0 < nagg; n0 += 2) {
        const int n = n0 + sel;
        const bool has = lane_on && n < nagg;
        const uint32_t qa = member(n0), qb = member(min(n0 + 1, nagg - 1));
        const uint32_t q = sel ? qb : qa;
        const int qx = nlk_x(q), qy = nlk_y(q);
        if (has) nlk_load_row8(src_c + qy * g.w + qx, x);
        nlk_dct8x8_fwd(x, ck);
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = gain[r] * x[r] + mu[r];  // mu already holds (1 - a) * M
        nlk_dct8x8_inv(x, cik);
        // the two members of a step may overlap: update the tile one after the other
        if (has && sel == 0) add_patch(qx, qy, x);
        if (has && sel == 1) add_patch(qx, qy, x);
      }
    } else {
      for (int n = 0; n < nagg; ++n) {
        const uint32_t q = member(n);
        const int qx = nlk_x(q), qy = nlk_y(q);
        const bool ld = lane_on && (sel == 0 || !passthrough);
        if (ld) nlk_load_row8((sel ? prev_c : src_c) + qy * g.w + qx, x);
        if (!passthrough) {
          nlk_dct8x8_fwd(x, ck);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float y0 = nlk_dpp<NLK_DPP_ROR8>(x[r]);  // previous-frame coefficient
            x[r] = (1 - gain[r]) * x[r] + gain[r] * y0;    // reference: :1775
          }
          nlk_dct8x8_inv(x, cik);
        }
        if (lane_on && sel == 0) add_patch(qx, qy, x);
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
