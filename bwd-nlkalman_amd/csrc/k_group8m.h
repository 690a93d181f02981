// k_group8m.h — group processing for 8x8 patches on the matrix cores
// (reference: src/nlkalman.c:713-932 filter, :1603-1845 smoother).
//
// The 2-D DCT of an 8x8 patch is folded by the even/odd symmetry of the DCT-II
// basis in both directions: with F_q[i][j] = X[i][j] +- X[i][7-j] +- X[7-i][j]
// +- X[7-i][7-j] (q = parity of the vertical / horizontal frequency, i, j < 4)
//   Y[2a+qr][2b+qc] = sum_{i,j} C[2a+qr][i] C[2b+qc][j] F_q[i][j]
// It runs in two forms, chosen per pass by the template argument SEP (tu_group8.hip picks by mode):
//   Kronecker  four 16x16 matrices D_q applied to four 16-vectors: 1024 MACs per patch, all of them
//              useful, as v_mfma_f32_16x16x4_f32 products over 16 patches at a time (exact f32: the
//              f32 MFMA is an fmaf chain); the inverse uses the same matrices. Candidates land in
//              REGISTERS: what the statistics of pass A want. Roles below.
//   separable  Y_q = C_qr F_q C_qc^T, 512 MACs per patch, on v_mfma_f32_4x4x1_16B_f32 with one patch
//              per lane quad (round 5; described where its helpers are defined): what pass B runs
//              for the first iteration's groups; a separable pass A exists (SEP bit 2) and loses.
// Both unfold with the same butterflies.
//
// Lane roles of the Kronecker form (lo = lane & 15, g4 = lane >> 4):
//   loads    lane = patch slot lo, rows g4 and 7-g4 of that patch (2 x 8 floats);
//            its folded values F_q[g4][s] are the MFMA operand of k-step s.
//   pass A   C = X^T D_q^T: register j of quadrant q = coefficient lo of
//            candidate 4*g4+j. Statistics over the candidates are therefore
//            register sums plus one reduction over the four lane groups; the
//            gains come out per coefficient along the lanes.
//   pass B   slot = 4*channel + member. C' = D_q X (operands swapped) leaves
//            coefficient 4*g4+j of slot lo in register j, which is exactly the
//            operand layout of the inverse product X^T = Y^T D_q; its result has
//            the folded pixel lo of slot 4*g4+j in register j: lane group =
//            channel plane, register = member.
// Aggregation: after unfolding, lane (pixel lo, plane g4) holds 4 pixels of each
// of the 4 members of a step; one LDS read-modify-write per (member, pixel)
// touches 16 distinct pixels in (CH + 1) distinct planes (the weight plane is
// produced by the same path: gain 0 and the DCT of a constant-1 patch as mean),
// so the private accumulator tile needs no atomics (k_group8.h explains the
// tile and its flush).
#pragma once
#include "nlk_common.h"
#include "k_commit_rows.h"
#include "k_group8.h"
#include <type_traits>

typedef float nlk_f4 __attribute__((ext_vector_type(4)));

// tiles per chunk of the XCD-aware tile order (see the kernel); the grid is nlk_g8m_grid(ntx, nty) workgroups
// floats per channel of the gain stash ([gain | (1-gain)*mean][quadrant][16 coefficients] = 128, + 8: pass B reads
// 16 bytes per lane (a ds_read_b128: groups of 16 lanes on 64 banks, MI355X_MICROARCH.md) at channel * stride + 4 * lane group, and with a stride of 128 = 0 (mod 64 banks) two channels
// of every 16-lane group met on one bank: round 2's 19 % SQ_LDS_BANK_CONFLICT, now 0. Not + 16: 12.9 KB instead of
// 12.7 KB per workgroup costs a workgroup per CU - LDS is handed out in 1280-byte pieces - and 3.7 % of the time)
#define NLK_G8_SST 136
#ifndef NLK_G8_CW
#define NLK_G8_CW 16
#define NLK_G8_CH 4
#endif
static inline int nlk_g8m_grid(int ntx, int nty) {
  const int nch = ((ntx + NLK_G8_CW - 1) / NLK_G8_CW) * ((nty + NLK_G8_CH - 1) / NLK_G8_CH);
  return ((nch + 7) / 8) * 8 * NLK_G8_CW * NLK_G8_CH;
}

__device__ __forceinline__ float nlk_bperm(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ uint32_t nlk_bperm_u(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// rows g4 and 7-g4 of the 8x8 patch at p (row stride w). Always executed: slots without
// a patch are pointed at some valid patch by the caller and masked out later (loads under
// divergent control flow would make the compiler wait for every load in flight).
// ... the same rows addressed as ONE base pointer (wave-uniform: scalar registers) + 32-bit byte offsets per lane:
// the loads take the "scalar base + vector offset" form and a step's address arithmetic is two 32-bit adds
// instead of five 64-bit ones. `e` = element offset of the patch from `base` (image and channel plane included),
// rowa / rowb = g4 * w and (7 - g4) * w. Every image of a frame call lives in one allocation of less than 4 GiB
// (nlk_hip.hip: plan_frame), so a byte offset fits 32 bits.
__device__ __forceinline__ void nlk_rows_load32(const float* __restrict__ base, uint32_t e, uint32_t rowa, uint32_t rowb,
                                                float (&R)[16]) {
  typedef const __attribute__((address_space(1))) nlk_f4u* gp4;
  const char* b = reinterpret_cast<const char*>(base);
  const uint32_t oa = (e + rowa) * 4u, ob = (e + rowb) * 4u;
  const nlk_f4u a0 = *(gp4)(b + (size_t)oa), a1 = *(gp4)(b + (size_t)oa + 16), b0 = *(gp4)(b + (size_t)ob), b1 = *(gp4)(b + (size_t)ob + 16);
#pragma unroll
  for (int c = 0; c < 4; ++c) { R[c] = a0[c]; R[4 + c] = a1[c]; R[8 + c] = b0[c]; R[12 + c] = b1[c]; }
}

__device__ __forceinline__ void nlk_rows_load(const float* __restrict__ p, int w, int g4,
                                              float (&R)[16]) {
  // (explicit global address space: a pointer selected at run time would otherwise be
  // loaded with flat instructions, which also count on the LDS counter)
  typedef const __attribute__((address_space(1))) nlk_f4u* gp4;
  const float* ra = p + g4 * w;
  const float* rb = p + (7 - g4) * w;
  const nlk_f4u a0 = *(gp4)(ra), a1 = *(gp4)(ra + 4), b0 = *(gp4)(rb), b1 = *(gp4)(rb + 4);
#pragma unroll
  for (int c = 0; c < 4; ++c) { R[c] = a0[c]; R[4 + c] = a1[c]; R[8 + c] = b0[c]; R[12 + c] = b1[c]; }
}

// the two rows -> folded values F[q][s] (q = 2*qr + qc, see the header)
__device__ __forceinline__ void nlk_fold(const float (&R)[16], float (&F)[4][4]) {
  float P[8], M[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { P[c] = R[c] + R[8 + c]; M[c] = R[c] - R[8 + c]; }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    F[0][s] = P[s] + P[7 - s];
    F[1][s] = P[s] - P[7 - s];
    F[2][s] = M[s] + M[7 - s];
    F[3][s] = M[s] - M[7 - s];
  }
}

// the 16 folded values and an offset through one empty asm statement: whatever is computed from `o` afterwards
// (the addresses of the next loads) cannot be issued before F exists
#define NLK_PIN_FOLD(F, o)                                                                                        \
  asm volatile("" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[0][2]), "+v"(F[0][3]), "+v"(F[1][0]), "+v"(F[1][1]),    \
               "+v"(F[1][2]), "+v"(F[1][3]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[2][2]), "+v"(F[2][3]),          \
               "+v"(F[3][0]), "+v"(F[3][1]), "+v"(F[3][2]), "+v"(F[3][3]), "+v"(o))

// One quadrant's 16 x 16 product: C += X D (the data x(s = 0..3) as the A operand: its rows are the lanes' slots)
// or, SWAP, C += D X (the data as the B operand: its columns are the lanes' slots). Exact f32: the f32 MFMA is an
// fmaf chain. (Round 4 tried the f16 matrix cores at f32 accuracy instead - every operand split in two f16 halves,
// x = x1 + x2, d = d1 + d2, two v_mfma_f32_16x16x32_f16 with operands [x1 | x1] / [x2 | x2] against [d1 | d2] for the
// four f32 MFMAs, products exact in f32, round-trip error 4e-4 against 1.5e-4: parity green, but the ~16 vector
// instructions per quadrant that split the operands cost more than the 96 matrix-core cycles they save:
// group 0.866 ms against 0.832, profiles/README.md round 4.)
typedef nlk_f4 nlk_basis_op;   // d(s = 0..3) of a quadrant
template <bool SWAP>
__device__ __forceinline__ nlk_f4 nlk_mfma_q(const float (&x)[4], const nlk_basis_op& d, nlk_f4 C) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
    C = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(d[s], x[s], C, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], d[s], C, 0, 0, 0);
  return C;
}

// C_q += X^T D_q^T (patches along the rows) or D_q X (swapped: coefficients along the rows)
// PRIO: the wavefront's issue priority raised (s_setprio) over the run of matrix instructions, so that a wavefront
// whose products are ready goes before its neighbours' address arithmetic. Same box, group ms, off -> on (levels 1, 2,
// 3 alike): RGB FLT1 temporal 0.7215 -> 0.7150, first frame 0.961 -> 0.948, FLT2 0.613 -> 0.609, SMO1 1.291 -> 1.288;
// one channel FLT1 0.394 -> 0.390, first frame 0.480 -> 0.473, FLT2 unchanged, SMO1 0.547 -> 0.555 (that one stays
// without: PRIO in the kernel)
template <bool SWAP, bool PRIO>
__device__ __forceinline__ void nlk_mfma_fwd(const float (&F)[4][4], const nlk_basis_op (&dA)[4], nlk_f4 (&C)[4]) {
  if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int q = 0; q < 4; ++q) C[q] = nlk_mfma_q<SWAP>(F[q], dA[q], C[q]);
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}

// sum over the 64 lanes, in every lane, without LDS: two butterfly steps inside the quads and two mirror steps
// inside the rows of 16 lanes (DPP), then the two row-swap instructions (tools/ubench/permlane_swap.hip)
__device__ __forceinline__ float nlk_wave_sum_dpp(float v) {
  v += nlk_dpp<NLK_DPP_XOR1>(v);
  v += nlk_dpp<NLK_DPP_XOR2>(v);
  v += nlk_dpp<NLK_DPP_HMIRROR>(v);
  v += nlk_dpp<0x140 /* row_mirror */>(v);
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---- the separable form (round 5). Per parity quadrant Y_q = C_qr F_q C_qc^T with the 4 x 4
// even / odd halves of the basis: 2 x 64 MACs instead of the 256 of the Kronecker matrix D_q, on
// v_mfma_f32_4x4x1_16B_f32 - sixteen independent 4 x 4 blocks, one per lane quad, operand layout and exactness
// (a k = 1 step is one fmaf) checked by tools/ubench/mfma4x4.hip:
//     D[i][j] (register i of lane 4 b + j) += A[i] (lane 4 b + i) * B[j] (lane 4 b + j)      for every block b
// Lane = 4 * patch + folded row i; F_q[i][k] in register k. Stage 1, T = F_q C_qc^T: A = F_q[.][k], B = the lane
// constant C[2 j + qc][k] - its result T[i][j] sits in lane 4 p + j, register i, which IS the B operand of stage 2,
// Y = C_qr T (A = the lane constant C[2 a + qr][i]): no shuffle between the stages, and both stages use the same
// eight constants E[parity][k] = C[2 (lane & 3) + parity][k]. The inverse takes the coefficients as the A operand
// of its first stage (U^T = Y^T C_qr, B = G[qr][a] = C[2 a + qr][lane & 3]) and the result as the B operand of the
// second (A = G[qc][b]): X_q[i][m] comes out in lane 4 p + i, register m - the layout the rows were loaded in.
__device__ __forceinline__ nlk_f4 nlk_mfma4(float a, float b, nlk_f4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// Y[q] (register a: coefficient (2 a + qr, 2 (lane & 3) + qc) of the quad's patch) += forward transform of F[q]. The four
// quadrants together: the first stages of all of them, then the second stages - no MFMA waits for the one in front of
// it (C2 group 0.749 -> 0.742 ms against quadrant after quadrant)
template <bool PRIO>
__device__ __forceinline__ void nlk_sep_fwd4(const float (&F)[4][4], const float (&E)[2][4], nlk_f4 (&Y)[4]) {
  nlk_f4 T[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) T[q] = nlk_f4{0.f, 0.f, 0.f, 0.f};
  if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) T[q] = nlk_mfma4(F[q][k], E[q & 1][k], T[q]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) Y[q] = nlk_mfma4(E[q >> 1][i], T[q][i], Y[q]);
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}
// X[q] (register m: folded pixel (lane & 3, m) of the quad's patch) = inverse transform of Y[q]
template <bool PRIO>
__device__ __forceinline__ void nlk_sep_inv4(const nlk_f4 (&Y)[4], const float (&G)[2][4], nlk_f4 (&X)[4]) {
  nlk_f4 U[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { U[q] = nlk_f4{0.f, 0.f, 0.f, 0.f}; X[q] = nlk_f4{0.f, 0.f, 0.f, 0.f}; }
  if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) U[q] = nlk_mfma4(Y[q][a], G[q >> 1][a], U[q]);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) X[q] = nlk_mfma4(G[q & 1][b], U[q][b], X[q]);
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}
// floats per channel slot of the gain stash in the separable form: [gain | (1-gain)*mean][lane & 3][quadrant][a],
// + 4: a lane reads 16 bytes (ds_read_b128: 64 banks; the 4-byte tile updates see 32: tu_group8.hip) at slot * stride + 16 * (lane & 3) + 4 * q - with a stride of 4 (mod 64 banks) the
// (plane, lane & 3) pairs of a wavefront meet on no bank
#define NLK_G8S_SST 132
// stash slots of the separable form: the image channels, the weight plane, (1-channel frames) one all-zero slot
// for the idle planes; then 64 floats for -x0 (pass A)
template <int CH> constexpr int nlk_g8s_slots() { return CH + 2 < 4 ? CH + 2 : 4; }
// SEP: bit 1 = pass B in the separable form, bit 2 = pass A
template <int CH, int SEP> constexpr int nlk_g8_stash_floats() {
  return ((SEP & 2) ? nlk_g8s_slots<CH>() * NLK_G8S_SST : (CH + 2) * NLK_G8_SST) + ((SEP & 4) ? 64 : 0);
}

#ifndef NLK_G8_S5TRACK
#define NLK_G8_S5TRACK 1
#endif
#ifndef NLK_G8_WPS
#define NLK_G8_WPS 3  // wavefronts per SIMD the register budget is cut for (experiments: -DNLK_G8_WPS=2)
#endif
// (UNIT: which translation unit instantiated the kernel - the same source compiled under two instruction schedulers,
// tu_group8.hip and tu_group8_ilp.hip; it changes nothing but the symbol)
template <int CH, bool SMO, int SEP, int UNIT = 0>
__global__ void __launch_bounds__(64, NLK_G8_WPS)
k_group8m(const float* __restrict__ img,   // matching / statistics image (planar)
          const float* __restrict__ cur,   // image whose patches are filtered
          const float* __restrict__ prev,  // previous output or nullptr
          const uint8_t* __restrict__ vmap, NlkGeom g, NlkGTile tl,
          const uint32_t* __restrict__ topk, const NlkTarget* __restrict__ tinfo,
          const uint32_t* __restrict__ gcoords, const uint8_t* __restrict__ active,
          const float* __restrict__ basis,   // [8][8] orthonormal DCT-II
          const float* __restrict__ window,  // [8][8] aggregation window
          float* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [(CH+1)][plane]
  constexpr int PSZ = 8, step = 4;
  const int lane = threadIdx.x;
  // Tile order. Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, nlk_common.h); tiles that
  // share halo rows should meet in one XCD's L2, but one contiguous band of the frame per XCD (nlk_xcd_tile)
  // leaves whole XCDs idle at the end of a launch - the bands differ in skipped targets (2 % of the kernel) -
  // and plain launch order spreads every neighbourhood over all eight L2s (8x the fetched bytes). So: chunks
  // of NLK_G8_CW x NLK_G8_CH tiles, dealt round-robin to the XCDs, each worked through by one XCD.
  uint32_t chase_gen = 0;
  if (tl.chase) chase_gen = *tl.chase_gen;  // (advanced by the bit-plane kernel in front of this launch)
  // the replay of the grid rows [0, nrows) from the bit planes, decisions published as tagged words (k_commit_rows.h)
  auto chase_replay = [&](int nrows) {
    __builtin_amdgcn_s_setprio(3);
    if (tl.chase == 1)
      nlk_commit_rows1<8, true>(tl.chase_planes, nullptr, nullptr, tl.chase_words, chase_gen, g.ngx, 0, nrows, lane);
    else if (tl.chase == 2)
      nlk_commit_rows<2, true>(tl.chase_planes, nullptr, nullptr, tl.chase_words, chase_gen, g.ngx, 0, nrows, lane);
    else
      nlk_commit_rows<3, true>(tl.chase_planes, nullptr, nullptr, tl.chase_words, chase_gen, g.ngx, 0, nrows, lane);
    __builtin_amdgcn_s_setprio(0);
  };
  if (tl.chase && blockIdx.x == 0 && !tl.chase_test_skip0) {
    // the launch's first workgroup replays the processed mask of the grid rows down to the launch's last one before
    // its own tile; every workgroup, this one included, then waits for the decision words of its targets below
    chase_replay(tl.chase_rows);
  }
  int tile_x, tile_y, gx0, gy0, cx, cy;
  if ((int)blockIdx.x < tl.nmain) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int nchx = (tl.ntx + NLK_G8_CW - 1) / NLK_G8_CW;
    const int chunk = (i / (NLK_G8_CW * NLK_G8_CH)) * 8 + xcd, within = i % (NLK_G8_CW * NLK_G8_CH);
    tile_x = (chunk % nchx) * NLK_G8_CW + within % NLK_G8_CW;
    tile_y = (chunk / nchx) * NLK_G8_CH + within / NLK_G8_CW;
    if (tile_x >= tl.ntx || tile_y >= tl.nty) return;
    // The launch ends when its last tile does, and a tile of 3 x 2 targets takes 1/7 of a 1080p launch: the last
    // tile rows hold one grid row instead of tgy, and the very last grid rows go one target per workgroup
    // (tu_group8.hip; deterministic mode keeps the uniform tiles its gather kernel assumes).
    const bool full = tile_y < tl.nty_full;
    gx0 = tile_x * tl.tgx;
    gy0 = full ? tile_y * tl.tgy : tl.nty_full * tl.tgy + (tile_y - tl.nty_full);
    cx = min(tl.tgx, g.ngx - gx0);
    cy = min(full ? tl.tgy : 1, g.ngy - tl.single - gy0);
  } else {
    // single targets: every XCD takes a contiguous run of them
    const int j = blockIdx.x - tl.nmain, n = tl.single * g.ngx, per = (n + 7) >> 3;
    const int idx = (j & 7) * per + (j >> 3);
    if ((j >> 3) >= per || idx >= n) return;
    tile_y = idx / g.ngx; tile_x = idx - tile_y * g.ngx;
    gx0 = tile_x; gy0 = g.ngy - tl.single + tile_y; cx = 1; cy = 1;
  }
  const int tile_id = tile_y * tl.ntx + tile_x;
  const int rx0 = max(gx0 * step - tl.wmax, 0);
  const int rx1 = min((gx0 + cx - 1) * step + tl.wmax + PSZ, g.w);
  const int ry0 = max(g.oy + gy0 * step - tl.wmax, 0);
  const int ry1 = min(g.oy + (gy0 + cy - 1) * step + tl.wmax + PSZ, g.h);
  const int rw = rx1 - rx0, rh = ry1 - ry0;
  const int rwp = tl.rwp, plane = tl.plane;
  // the tile's target records (requested first: they arrive while the tile is cleared)
  int rec_act = 0, rec_nsel = 0, rec_nagg = 0, chase_bit = 0;
  const uint64_t* chase_wp = tl.chase_words;
  uint64_t chase_v = 0;
  bool chase_late = false;
  uint32_t rec_vb[4] = {0u, 0u, 0u, 0u};
  if (lane < cx * cy) {
    const int ty = lane / cx, tx = lane - ty * cx;
    const size_t t = (size_t)(gy0 + ty) * g.ngx + gx0 + tx;
    if (tl.chase) {
      // decisions arrive as generation-tagged words (indivisible 64-bit stores / loads at agent scope: no fence, no
      // stale line of another XCD's L2). The replay runs ~30x faster than the launch walks through the grid rows:
      // only the first wave of workgroups ever waits - a few microseconds. HIP promises nothing about the order
      // workgroups are dispatched in, so a workgroup that has waited ~80 us (three times what the whole replay
      // takes) stops waiting for workgroup 0 and replays the rows it needs itself, below: the same words with the
      // same generation, idempotent, and whoever else is waiting for them sees them too.
      chase_wp = tl.chase_words + (size_t)(tl.chase_row0 + gy0 + ty) * 64 + ((gx0 + tx) >> 5);
      chase_v = __hip_atomic_load(chase_wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the wait grows with the row: the replay reaches row r after ~r x 90 ns, a spin is ~1.2 us - on an 8K-class grid a
      // fixed 64 spins would make every first-wave workgroup of the lower rows replay for itself: ADVICE r5)
      const int spin_max = 64 + ((tl.chase_row0 + gy0 + ty) >> 2);
      for (int spin = 0; (uint32_t)(chase_v >> 32) != chase_gen && spin < spin_max; ++spin) {
        __builtin_amdgcn_s_sleep(16);
        chase_v = __hip_atomic_load(chase_wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      chase_late = (uint32_t)(chase_v >> 32) != chase_gen;
      chase_bit = (gx0 + tx) & 31;
    } else {
      rec_act = active[t];
    }
    const NlkTarget info = tinfo[t];
    rec_nsel = info.nsel; rec_nagg = info.nagg;
    rec_vb[0] = (uint32_t)info.vbits[0]; rec_vb[1] = (uint32_t)(info.vbits[0] >> 32);
    rec_vb[2] = (uint32_t)info.vbits[1]; rec_vb[3] = (uint32_t)(info.vbits[1] >> 32);
  }
  if (tl.chase) {
    if (__ballot(chase_late)) {  // (wave-uniform) self-rescue: the rows down to this tile's last one
      chase_replay(tl.chase_row0 + gy0 + cy);
      // (the words were stored by other lanes of this wavefront: release after the replay, acquire on the reload -
      // a formal happens-before instead of relying on the in-order issue of the vector memory instructions)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane < cx * cy) chase_v = __hip_atomic_load(chase_wp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane < cx * cy) rec_act = ((uint32_t)chase_v >> chase_bit) & 1u;
  }
  if (tl.split) {  // (the two launches of deterministic mode: most tiles of the far one have nothing to do)
    const bool work = rec_act && rec_nagg != 0 &&
                      nlk_far_target(g, (rec_vb[0] | rec_vb[1] | rec_vb[2] | rec_vb[3]) ? 1 : 0) == (tl.far != 0);
    if (!__ballot(work)) {
      if (threadIdx.x == 0) tl.tflag[tile_id] = 0;
      return;
    }
  }
  bool any_target = false;  // (deterministic mode: a tile without work writes no slab)
  for (int i = lane; i < (CH + 1) * plane / 4; i += 64)  // (plane is a multiple of 16)
    reinterpret_cast<nlk_f4*>(smem)[i] = nlk_f4{0.f, 0.f, 0.f, 0.f};
  // [CH+2][NLK_G8_SST]: gains / means between the passes ([2][4][16] floats per channel + padding). Pass B treats the weight plane as one more
  // channel (gain 0, mean = DCT of a constant-1 patch: 8 at the DC coefficient) and the unused slots
  // of a 1-channel frame as another (all zero), so that its shrinkage is one fma without selects
  float* stash = smem + (CH + 1) * plane;
  // which pass runs the separable form (bit 1: pass B; bit 2: pass A)
  constexpr bool SEPA = (SEP & 4) != 0, SEPB = (SEP & 2) != 0;
  constexpr bool PRIO = !(CH == 1 && SMO);  // (raised issue priority over the matrix instructions: nlk_mfma_fwd)
  // single-channel frames in the separable pass B: SIXTEEN members per step (below) instead of 4 members x {image,
  // weights, idle, idle}
  constexpr bool G16 = SEPB && CH == 1;
  constexpr int SST = SEPB ? NLK_G8S_SST : NLK_G8_SST;      // floats per stash slot
  constexpr int NSLOT = SEPB ? nlk_g8s_slots<CH>() : CH + 2;
  for (int i = lane; i < (NSLOT - CH) * SST; i += 64) stash[CH * SST + i] = (i == 64) ? 8.f : 0.f;
  __syncthreads();

  const int lo = lane & 15, g4 = lane >> 4;
  // separable form: lane = 4 * patch slot + folded row; in pass B a step's slot = 4 * member + plane
  const int si = lane & 3, sp = lane >> 2, spl = sp & 3;
  // Kronecker form: D_q as the forward operand (dA: D_q[coef lo][pixel 4*g4+s]) and as the inverse
  // operand (dI: D_q[coef 4*g4+s][pixel lo]); q = 2*qr + qc
  nlk_basis_op dA[4], dI[4];
  // separable form: E[parity][k] = C[2 si + parity][k] (forward, both stages), G[parity][a] = C[2 a + parity][si]
  float sE[2][4], sG[2][4];
  if constexpr (SEP != 0) {
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sE[par][s] = basis[(2 * si + par) * 8 + s];
        sG[par][s] = basis[(2 * s + par) * 8 + si];
      }
  }
  // Hybrid form (SEP == 2): the Kronecker operand of pass A is the product of two basis values,
  // dA[q][s] = C[2 (lo >> 2) + qr][g4] * C[2 (lo & 3) + qc][s] = Er[qr] * sE[qc][s] (lo & 3 == si), so only Er is
  // kept across the targets and the sixteen products are rebuilt where a target's pass A starts: pass B, which
  // needs its registers for the second PX buffer of its software pipeline, does not carry them.
  float Er[2] = {0.f, 0.f};
  if constexpr (SEP == 2) {
    Er[0] = basis[(2 * (lo >> 2) + 0) * 8 + g4];
    Er[1] = basis[(2 * (lo >> 2) + 1) * 8 + g4];
  }
  if constexpr (SEP == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int qr = q >> 1, qc = q & 1;
      float a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = basis[(2 * (lo >> 2) + qr) * 8 + g4] * basis[(2 * (lo & 3) + qc) * 8 + s];
        b[s] = basis[(2 * g4 + qr) * 8 + (lo >> 2)] * basis[(2 * s + qc) * 8 + (lo & 3)];
      }
      dA[q] = nlk_f4{a[0], a[1], a[2], a[3]};
      dI[q] = nlk_f4{b[0], b[1], b[2], b[3]};
    }
  }
  // aggregation role. Kronecker form: folded pixel lo = (pi, pj) of plane g4 -> 4 pixels of the patch.
  // Separable form (after the transposition of pass B): plane spl, rows si and 7 - si, columns g4 and 4 + g4.
  const int pi = lo >> 2, pj = lo & 3;
  const int aplane = SEPB ? spl : g4;
  const bool agg_on = aplane <= CH;
  int poff[4];
  float win[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int r = SEPB ? ((kk & 2) ? 7 - si : si) : ((kk & 2) ? 7 - pi : pi);
    const int c = SEPB ? 4 * (kk & 1) + g4 : ((kk & 1) ? 7 - pj : pj);
    poff[kk] = (G16 ? 0 : (agg_on ? aplane : 0) * plane) + r * rwp + c;  // (G16: the plane is chosen per round)
    win[kk] = window[r * 8 + c];
  }
  const size_t npix = (size_t)g.w * g.h;
  const float* src = g.have_basic ? cur : img;  // patches that get filtered
  // the planar images as element offsets from one base (tl.pbase = the start of the context's image slab)
  const float* const pbase = tl.pbase;
  const uint32_t e_img = (uint32_t)(img - pbase), e_src = (uint32_t)(src - pbase), e_prev = prev ? (uint32_t)(prev - pbase) : e_img;
  // the two patch rows a lane loads: lrow and 7 - lrow (by pass)
  const int lrowA = g4, lrowB = SEPB ? si : g4;  // (the separable pass A loads one row per lane: there)
  const uint32_t rowa = (uint32_t)(lrowA * g.w), rowb = (uint32_t)((7 - lrowA) * g.w);
  const uint32_t rowaB = (uint32_t)(lrowB * g.w), rowbB = (uint32_t)((7 - lrowB) * g.w);
  const float s2 = g.sigma2;
  // pass-B role of the lane as a load slot: channel / member of its slot
  const int bch = SEPB ? spl : lo >> 2, bm = SEPB ? g4 : lo & 3;

  for (int tt = 0; tt < cx * cy; ++tt) {
    if (!__builtin_amdgcn_readlane(rec_act, tt)) continue;
    const int nagg = __builtin_amdgcn_readlane(rec_nagg, tt);
    if (nagg == 0) continue;
    if (tl.split) {  // (deterministic mode: near and far groups in separate launches)
      const uint32_t anyv = (uint32_t)__builtin_amdgcn_readlane((int)(rec_vb[0] | rec_vb[1] | rec_vb[2] | rec_vb[3]), tt);
      if (nlk_far_target(g, anyv ? 1 : 0) != (tl.far != 0)) continue;
    }
    any_target = true;
    const int ty = tt / cx, tx = tt - ty * cx;
    const size_t t = (size_t)(gy0 + ty) * g.ngx + gx0 + tx;
    const int k = __builtin_amdgcn_readlane(rec_nsel, tt);

    // candidate / member lists: one entry per lane (two rounds); validity and
    // group-membership bits of the candidates as wave-uniform masks
    uint32_t qreg[2], greg[2];
    uint64_t vbits[2], gbits[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int i = lane + 64 * m;
      qreg[m] = i < k ? topk[t * g.kmax + i] : 0u;
      greg[m] = i < nagg ? gcoords[t * g.gstride + i] : 0u;
    }
    // (validity bits of the kept candidates come with the records of the match kernel)
#pragma unroll
    for (int m = 0; m < 2; ++m)
      vbits[m] = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rec_vb[2 * m], tt) |
                 ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rec_vb[2 * m + 1], tt) << 32);
    const int np0a = __popcll(vbits[0]);
    const int np0 = np0a + __popcll(vbits[1]);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int rank = (m ? np0a : 0) + __popcll(vbits[m] & ((1ull << lane) - 1ull));
      gbits[m] = __ballot(((vbits[m] >> lane) & 1ull) && rank < g.ntagg);
    }
    const int np1 = k;
    const int ngrp = min(np0, g.ntagg);
    // (1 / n for n <= 128 by v_rcp_f32: within 1 ulp, like the gains' reciprocals)
    const float in1 = np1 ? __builtin_amdgcn_rcpf((float)np1) : 0.f;
    const float in0 = np0 ? __builtin_amdgcn_rcpf((float)np0) : 0.f;
    const float ing = ngrp ? __builtin_amdgcn_rcpf((float)ngrp) : 0.f;
    const bool passthrough = SMO && np0 == 0;  // reference: :1795-1804

    // ---------------- pass A: statistics over the k kept candidates, one channel at a
    // time. A step transforms 16 patches: 16 candidates of the image, or - when the
    // target has previous-frame patches - 8 candidates of the image AND their 8
    // previous-frame patches (slot lo = 4*(c>>1) + 2*isprev + (c&1), so that both
    // coefficients of a candidate end up in the same lane: registers j and j+2).
    //
    // Sums are taken of d = coefficient - x0 (x0 = coefficient of the first candidate's image
    // patch, which every MFMA chain starts from as its C operand: NX0 = -x0, no instruction):
    //   S0/S1 image (all candidates), S2/S3 previous frame (valid ones), S4 squared
    //   image-previous difference, S5 previous frame over the group members.
    // Slots that must not count - candidates past the k-th, and in the filter a candidate without
    // a valid previous patch (the Kalman branch uses no image statistics, reference: :859-904) -
    // read the FIRST candidate's image patch instead: their d is zero up to the rounding of the
    // chain (a few ulp of x0, against sums of ~k * sigma), so the sums need no masks. What still
    // needs one: the group mean (members only) and, in the smoother, the transition term of a
    // candidate whose image patch counts while its previous patch does not (reference: :1659-1667).
    float part_sum = 0.f;
    // per lane: offset (floats) of candidate (lane + 64 m)'s patch inside an image plane,
    // bit 31 = "has a valid previous patch"
    uint32_t oreg[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
      oreg[m] = (uint32_t)(nlk_y(qreg[m]) * g.w + nlk_x(qreg[m])) | ((uint32_t)((vbits[m] >> lane) & 1ull) << 31);
    const uint32_t o_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)oreg[0]) & 0x7fffffffu;
    if constexpr (!SEPA) {
    if constexpr (SEP == 2) {
      float e0 = Er[0], e1 = Er[1];
      asm volatile("" : "+v"(e0), "+v"(e1));  // (per target: the products must not be hoisted out of the target loop)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dA[q] = nlk_f4{(q >> 1 ? e1 : e0) * sE[q & 1][0], (q >> 1 ? e1 : e0) * sE[q & 1][1],
                       (q >> 1 ? e1 : e0) * sE[q & 1][2], (q >> 1 ? e1 : e0) * sE[q & 1][3]};
    }
    float S[6][4];
    nlk_f4 NX0[4];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) S[a][q] = 0.f;
    // MODE 0: no previous-frame patches (Wiener branch; 16 candidates per step), 1: filter with
    // previous-frame patches (Kalman branch), 2: smoother with previous-frame patches.
    // Schedule: the rows of step it+1 are requested right after those of step it were folded (into the
    // same registers) and the offset of step it+2 is prepared, so every request has a whole step to land.
    // All loads are unconditional: a load under a run-time branch would turn every wait into a wait for
    // all loads in flight.
    auto pass_a = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool HP = MODE != 0;
      constexpr int CB = HP ? 8 : 16;  // candidates per step
      const int nb = (k + CB - 1) / CB;
      if (nb == 0) return;
      const int slot_c = HP ? 2 * (lo >> 2) + (lo & 1) : lo;
      const bool slot_prev = HP && ((lo >> 1) & 1);
      const uint32_t live_need = MODE == 1 ? 0x80000000u : (slot_prev ? 0x80000000u : 0u);
      // slot lo's patch in step bb: offset into an image plane, and whether it is the previous frame's
      // (element offset from the base: the image's or - for a live previous-frame slot - the previous frame's plane 0)
      const uint32_t e_slot = slot_prev ? e_prev : e_img, e_dead = o_first + e_img;
      auto slot_off = [&](int bb) -> uint32_t {
        const int ci = CB * bb + slot_c, cl = min(ci, k - 1);
        const uint32_t oc = nlk_bperm_u(cl < 64 ? oreg[0] : oreg[1], cl & 63);
        const bool live = ci < k && (oc & live_need) == live_need;
        return live ? (oc & 0x7fffffffu) + e_slot : e_dead;
      };
      // rows g4 and 7-g4 of the patch at element offset `off` (plane 0) of channel cc
      bool rows_first = true;
      auto rows_read = [&](uint32_t off, int cc, float (&R)[16]) {
#ifdef NLK_G8_EXP_NOROWS   // (timing experiment: only the first step's rows are loaded - what any prefetch scheme could gain at most)
        if (!rows_first) { asm volatile("" : "+v"(R[0]), "+v"(R[5]), "+v"(R[10]), "+v"(R[15]) : "v"(off)); return; }
        rows_first = false;
#endif
        nlk_rows_load32(pbase, off + (uint32_t)cc * (uint32_t)npix, rowa, rowb, R);
      };
      float R[16], F[4][4];
#if NLK_G8_S5TRACK
      bool s5_track = true;
#endif
      const uint32_t o_step0 = slot_off(0);
      rows_read(o_step0, 0, R);
      uint32_t onext = slot_off(1);
      for (int ch = 0; ch < CH; ++ch)
      for (int b = 0; b < nb; ++b) {
        nlk_fold(R, F);
        // the next step's rows; after a channel's last step the next channel's first (after the very
        // last step: a harmless reload)
        const bool wrap = b + 1 == nb;
        uint32_t o_load = wrap ? o_step0 : onext;
        // The reload below reuses R (no second register set). A sched_barrier alone does not order it: the fold has no
        // side effects, so instruction selection placed it BEHIND the loads and paid 16 v_mov per step to keep the
        // rows alive. The empty statement makes the load address depend on it: the folded values exist before the
        // loads are issued.
        NLK_PIN_FOLD(F, o_load);
        __builtin_amdgcn_sched_barrier(0);
        rows_read(o_load, wrap ? min(ch + 1, CH - 1) : ch, R);
        onext = slot_off(wrap ? 1 : b + 2);
        __builtin_amdgcn_sched_barrier(0);
        nlk_f4 C[4];
        if (b == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) C[q] = nlk_f4{0.f, 0.f, 0.f, 0.f};
          nlk_mfma_fwd<false, PRIO>(F, dA, C);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x0 = nlk_bperm(C[q][0], lo);
            NX0[q] = nlk_f4{-x0, -x0, -x0, -x0};
#pragma unroll
            for (int j = 0; j < 4; ++j) C[q][j] -= x0;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) C[q] = NX0[q];
          nlk_mfma_fwd<false, PRIO>(F, dA, C);
        }
        if (HP) {
          // bit j of the lane group's pair of candidates: group membership (filter) / valid previous patch (smoother)
          const uint64_t mw = MODE == 1 ? (b < 8 ? gbits[0] : gbits[1]) : (b < 8 ? vbits[0] : vbits[1]);
          const uint32_t mbyte = (uint32_t)(mw >> (8 * (b & 7))) & 0xffu;
#if NLK_G8_S5TRACK
          // (filter) The group members are the FIRST valid candidates of the sorted list: while every candidate of every
          // step so far was a member, the members' sum S5 IS the sum S2 over all candidates (fmaf(1, d, s) = s + d, same
          // order: the same bits) and is not kept; the first step that is not all members copies it, and only steps that
          // hold a member add to it afterwards - for a temporal target of 30 candidates and 20 members one masked step of
          // four per channel instead of four.
          if (MODE == 1 && s5_track && mbyte != 0xffu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) S[5][q] = S[2][q];
            s5_track = false;
          }
          const bool s5_add = MODE == 1 && !s5_track && mbyte != 0u;  // (wave-uniform)
#else
          constexpr bool s5_add = MODE == 1;
#endif
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float mk = (float)((mbyte >> (2 * g4 + j)) & 1u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float di = C[q][j], d = C[q][j + 2];
              if (MODE == 2) {
                S[0][q] += di;
                S[1][q] = fmaf(di, di, S[1][q]);
              }
              S[2][q] += d;
              S[3][q] = fmaf(d, d, S[3][q]);
              const float df = di - d;  // reference: :769-783, smoother :1659-1667
              if (MODE == 2) S[4][q] = fmaf(mk * df, df, S[4][q]);
              else S[4][q] = fmaf(df, df, S[4][q]);
            }
          }
          if (s5_add) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float mk = (float)((mbyte >> (2 * g4 + j)) & 1u);
#pragma unroll
              for (int q = 0; q < 4; ++q) S[5][q] = fmaf(mk, C[q][j + 2], S[5][q]);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              S[0][q] += C[q][j];
              S[1][q] = fmaf(C[q][j], C[q][j], S[1][q]);
            }
        }
        if (b == nb - 1) {
#if NLK_G8_S5TRACK
          if (MODE == 1 && s5_track) {  // (every candidate was a member)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[5][q] = S[2][q];
          }
          s5_track = true;
#endif
          // The channel is complete: candidates are spread over the four lane groups. Each of the
          // sums is reduced over them with the row-swap instructions: two registers per
          // v_permlane32_swap + add, two of those per v_permlane16_swap + add, which leaves the total
          // of S[a][q] in lane group q (tools/ubench/permlane_swap.hip) - 6 operations per 4 registers
          // instead of 16, and the gains below are computed once per (quadrant, coefficient): lane
          // (lo, g4) owns coefficient lo of quadrant g4.
          float T[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            T[a] = 0.f;
            const bool used = MODE == 0 ? a < 2 : (MODE == 1 ? a >= 2 : a < 5);
            if (used) {
              const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(S[a][0]), __float_as_uint(S[a][2]), false, false);
              const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(S[a][1]), __float_as_uint(S[a][3]), false, false);
              const float X = __uint_as_float(x[0]) + __uint_as_float(x[1]);
              const float Y = __uint_as_float(y[0]) + __uint_as_float(y[1]);
              const auto z = __builtin_amdgcn_permlane16_swap(__float_as_uint(X), __float_as_uint(Y), false, false);
              T[a] = __uint_as_float(z[0]) + __uint_as_float(z[1]);
            }
          }
          // ---- gain of coefficient lo of quadrant g4 (reference: :799-811, :859-904; smoother :1683-1776)
          {
            const float nx0q = g4 == 0 ? NX0[0][0] : (g4 == 1 ? NX0[1][0] : (g4 == 2 ? NX0[2][0] : NX0[3][0]));
            const float v0 = (T[3] - T[2] * T[2] * in0) * in0;  // previous-frame variance
            const float v01n = T[4] * in0;
            float a, term, m;
            if (MODE == 2) {
              const float v1 = (T[1] - T[0] * T[0] * in1) * in1;  // image variance
              a = v1 * __builtin_amdgcn_rcpf(v1 + g.beta_t * v01n);
              const float pv = v0 - g.beta_t * v01n;
              term = (1 - a * a) * v1 + a * a * (pv > 0.f ? pv : 0.f);
              m = 0.f;
            } else if (MODE == 1) {
              const float d = v01n - (g.have_basic ? 0.f : s2);
              const float v = v0 + (0.f > d ? 0.f : d);
              a = v * __builtin_amdgcn_rcpf(v + g.beta_t * s2);
              term = (1 - a * a) * v + a * a * s2;
              m = T[5] * ing - nx0q;
            } else {
              const float v1 = (T[1] - T[0] * T[0] * in1) * in1;
              const float d = v1 - (g.have_basic ? 0.f : s2);
              const float v = 0.f > d ? 0.f : d;
              a = v * __builtin_amdgcn_rcpf(v + g.beta_x * s2);
              term = a * v;
              m = T[0] * in1 - nx0q;
            }
            part_sum += term;
            // parked in LDS for pass B: [channel][gain | (1-a)*mean][quadrant][coefficient]
            // (filter: a*PG + (1-a)*M, reference: :879, :902)
            // (Kronecker pass B: [quadrant][coefficient]; separable pass B: [horizontal index][quadrant][vertical index])
            const int cidx = SEPB ? 16 * (lo & 3) + 4 * g4 + (lo >> 2) : 16 * g4 + lo;
            stash[ch * SST + cidx] = a;
            stash[ch * SST + 64 + cidx] = (1 - a) * m;
          }
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[a][q] = 0.f;
        }
      }
    };
    if (np0 == 0) pass_a(std::integral_constant<int, 0>{});
    else if (!SMO) pass_a(std::integral_constant<int, 1>{});
    else pass_a(std::integral_constant<int, 2>{});
    } else {
    // ---------------- pass A, separable form (SEP bit 2). The transforms leave a PATCH per lane quad and the
    // coefficients of a patch over the quad's lanes and registers, so the sums over the candidates are kept per lane
    // over the batches and reduced over the lane quads once per channel; lane (quadrant g4, vertical index spl,
    // horizontal index si) then owns one coefficient for the gains. The shift -x0 (the first candidate's image
    // coefficients: slot 0 of the first batch) is parked in LDS and is the C operand of every later chain; slots
    // that must not count read the first candidate's image patch, exactly as above.
    float* const x0buf = stash + NSLOT * SST;
    // gain of the coefficient a lane owns after the reduction over the lane quads - quadrant g4, vertical index spl,
    // horizontal index si - from the channel's totals T (reference: :799-811, :859-904; smoother :1683-1776)
    auto sep_gain = [&](auto mode_tag, const auto& T, int ch) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr int NS = MODE == 0 ? 2 : (MODE == 1 ? 4 : 5);
        // the coefficient this lane owns after the reduction below: quadrant oq = 2 qr + qc, vertical index oa,
        // horizontal index si (lane = 32 qc + 8 oa + 4 qr + si)
        const int oq = 2 * ((lane >> 2) & 1) + (lane >> 5), oa = (lane >> 3) & 3;
        const float nx0q = x0buf[16 * si + 4 * oq + oa];
        // (where pass B reads the gains: separable [horizontal index][quadrant][vertical index], Kronecker
        // [quadrant][coefficient])
        const int cidx = SEPB ? 16 * si + 4 * oq + oa : 16 * oq + 4 * oa + si;
        float a, term, m;
        if (MODE == 2) {
          const float v1 = (T[1] - T[0] * T[0] * in1) * in1;  // image variance
          const float v0 = (T[NS > 3 ? 3 : 0] - T[NS > 2 ? 2 : 0] * T[NS > 2 ? 2 : 0] * in0) * in0;  // previous-frame variance
          const float v01n = T[NS > 4 ? 4 : 0] * in0;
          a = v1 * __builtin_amdgcn_rcpf(v1 + g.beta_t * v01n);
          const float pv = v0 - g.beta_t * v01n;
          term = (1 - a * a) * v1 + a * a * (pv > 0.f ? pv : 0.f);
          m = 0.f;
        } else if (MODE == 1) {
          const float v0 = (T[1] - T[0] * T[0] * in0) * in0;
          const float v01n = T[NS > 2 ? 2 : 0] * in0;
          const float d = v01n - (g.have_basic ? 0.f : s2);
          const float v = v0 + (0.f > d ? 0.f : d);
          a = v * __builtin_amdgcn_rcpf(v + g.beta_t * s2);
          term = (1 - a * a) * v + a * a * s2;
          m = (T[0] - T[NS > 3 ? 3 : 0]) * ing - nx0q;
        } else {
          const float v1 = (T[1] - T[0] * T[0] * in1) * in1;
          const float d = v1 - (g.have_basic ? 0.f : s2);
          const float v = 0.f > d ? 0.f : d;
          a = v * __builtin_amdgcn_rcpf(v + g.beta_x * s2);
          term = a * v;
          m = T[0] * in1 - nx0q;
        }
        part_sum += term;
        // parked in LDS for pass B: [channel][gain | (1-a)*mean][si][quadrant][a]
        stash[ch * SST + cidx] = a;
        stash[ch * SST + 64 + cidx] = (1 - a) * m;
    };
    // HALF a patch per lane quad: lane = 8 slot + u OWNS row u = 0..7 of its candidate (it arrives as two half rows,
    // one loaded by the lane itself and one by its neighbour u ^ 1, so that every load touches one cache line per
    // lane pair: row_read / fold_half below), takes the mirror row 7 - u from lane u ^ 7 of its group of eight (ONE
    // DPP move, row_half_mirror: round 6; round 5 had the two halves in the two halves of the wavefront and fetched
    // the mirror row through ds_bpermute - 16 LDS round trips in every batch's dependency chain, whose issue is half
    // a Kronecker step's but which took longer) and keeps the sum P (u < 4: the quadrants of even vertical frequency)
    // or the difference M (u >= 4: odd; its quad holds the folded rows 3..0, so the second stage's constants are
    // reversed there). A batch = 8 candidates; a lane holds 2 quadrants x 4 vertical indices = 8 coefficients, so a
    // statistic is 8 registers (with a whole patch per quad - 16 candidates per batch, 16 registers per statistic -
    // the kernel spilled 27-65 registers and was slower: profiles/README.md round 5), both images' rows of the next
    // batch are requested a whole batch ahead, and the reduction over the 8 candidates' lanes is a reduce-scatter:
    // one row-swap level per register pair for lane bits 5 and 4, a rotation for bit 3 - 15 operations per
    // statistic, every lane ends up with the total of the ONE coefficient it owns.
    auto pass_a = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool HP = MODE != 0;
      constexpr int NS = MODE == 0 ? 2 : (MODE == 1 ? 4 : 5);
      const int nb = (k + 7) >> 3;
      if (nb == 0) return;
      const int hh = (lane >> 2) & 1, hp = lane >> 3;
      const bool odd = (lane & 1) != 0;
      // byte offsets of the two loads inside the candidate's patch (see row_read)
      const uint32_t rowx = (uint32_t)(((lane & 6) * g.w + (odd ? 4 : 0)) * 4);
      const uint32_t rowy = (uint32_t)((((lane & 6) + 1) * g.w + (odd ? 0 : 4)) * 4);
      const float sgn = hh ? -1.f : 1.f;
      float sEh[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sEh[i] = hh ? -sE[1][3 - i] : sE[0][i];  // (minus: fold_half leaves -M in the odd quads)
      const uint32_t e_dead = o_first + e_img;
      auto slot_offs = [&](int bb, uint32_t& oi, uint32_t& op) {
        const int ci = 8 * bb + hp, cl = min(ci, k - 1);
        const uint32_t oc = nlk_bperm_u(cl < 64 ? oreg[0] : oreg[1], cl & 63);
        const bool in_k = ci < k, valid = (oc >> 31) != 0u;
        const uint32_t o = oc & 0x7fffffffu;
        const bool live_i = MODE == 1 ? (in_k && valid) : in_k;
        oi = live_i ? o + e_img : e_dead;
        op = (in_k && valid) ? o + e_prev : e_dead;
      };
      int rows_seen = 0;
      auto row_read = [&](uint32_t off, int cc, float (&R)[8]) {
#ifdef NLK_G8_EXP_NOROWS   // (timing experiment: only the first batch's rows are loaded)
        if (rows_seen >= 2) { asm volatile("" : "+v"(R[0]), "+v"(R[3]), "+v"(R[5]), "+v"(R[7]) : "v"(off)); return; }
        ++rows_seen;
#endif
#ifdef NLK_G8_EXP_LDSROWS   // (timing experiment: the rows read from LDS - garbage from the tile - as a staged window would give them)
        if (rows_seen >= 2) {
          const float* wp = smem + ((off + (uint32_t)cc * 7u + (uint32_t)(lane & 7) * 18u) % 1500u);
#pragma unroll
          for (int c = 0; c < 8; ++c) R[c] = wp[c];
          return;
        }
        ++rows_seen;
#endif
        // lanes 2 m and 2 m + 1 own rows 2 m and 2 m + 1 of their candidate; each of the two loads takes ONE of those
        // rows, a half per lane, so that the two lanes read one cache line (the vector L1 takes a quad of lanes per
        // cycle and per line it touches: a row per lane - four lines per quad - cost twice the cycles of the
        // transforms themselves, profiles/README.md round 6): the first load row 2 m (even lane: columns 0..3, odd
        // lane: 4..7), the second row 2 m + 1 (even lane: 4..7, odd lane: 0..3). A lane then holds the LEFT half of
        // its own row - in R[0..3] (even) or R[4..7] (odd) - and its neighbour's right half; fold_half trades.
        typedef const __attribute__((address_space(1))) nlk_f4u* gp4;
        const char* bp = reinterpret_cast<const char*>(pbase);
        const uint32_t o = (off + (uint32_t)cc * (uint32_t)npix) * 4u;
        const nlk_f4u a0 = *(gp4)(bp + (size_t)(o + rowx)), a1 = *(gp4)(bp + (size_t)(o + rowy));
#pragma unroll
        for (int c = 0; c < 4; ++c) { R[c] = a0[c]; R[4 + c] = a1[c]; }
      };
      // A lane's row: left half Z from its own loads, right half from the neighbour lane (quad_perm [1, 0, 3, 2],
      // fused into the horizontal fold H = left[c] +- right[3 - c]); then the mirror row 7 - u from lane u ^ 7 of the
      // group of eight (row_half_mirror, fused into the vertical fold) -> P or M -> F[qc][s]
      auto fold_half = [&](const float (&R)[8], float (&F)[2][4]) {
        float Z[4], S[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          Z[c] = odd ? R[4 + c] : R[c];  // left half of the own row
          S[c] = odd ? R[c] : R[4 + c];  // what the neighbour needs: the right half of ITS row
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float rt = nlk_dpp<NLK_DPP_XOR1>(S[3 - c]);  // own row, column 7 - c
          F[0][c] = Z[c] + rt;
          F[1][c] = Z[c] - rt;
        }
        // own + mirror (u < 4) | own - mirror (u >= 4: MINUS the difference top - bottom; the sign is in the second
        // stage's constants sEh): F += sgn * F(lane u ^ 7), the mirror row as the DPP multiplicand of ONE v_fmac_f32
        // each. (The compiler keeps a v_mov_b32_dpp + v_fmac_f32 pair per value - its DPP combiner does not fold into
        // an accumulating operand -, 16 more vector instructions per batch; hence by hand. A DPP operand needs two
        // wait states after the instruction that wrote it, which the compiler does not insert for inline assembly: the
        // s_nop.)
#define NLK_G8_FMAC_HM(i) "v_fmac_f32_dpp %" #i ", %" #i ", %8 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        asm("s_nop 1\n\t" NLK_G8_FMAC_HM(0) NLK_G8_FMAC_HM(1) NLK_G8_FMAC_HM(2) NLK_G8_FMAC_HM(3) NLK_G8_FMAC_HM(4)
            NLK_G8_FMAC_HM(5) NLK_G8_FMAC_HM(6) NLK_G8_FMAC_HM(7)
            : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[0][2]), "+v"(F[0][3]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[1][2]),
              "+v"(F[1][3])
            : "v"(sgn));
#undef NLK_G8_FMAC_HM
      };
      auto fwd_half = [&](const float (&F)[2][4], nlk_f4 (&Y)[2]) {
        nlk_f4 T[2] = {nlk_f4{0.f, 0.f, 0.f, 0.f}, nlk_f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int qc = 0; qc < 2; ++qc) T[qc] = nlk_mfma4(F[qc][kk], sE[qc][kk], T[qc]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int qc = 0; qc < 2; ++qc) Y[qc] = nlk_mfma4(sEh[i], T[qc][i], Y[qc]);
      };
      nlk_f4* const x0w = reinterpret_cast<nlk_f4*>(x0buf + 16 * si + 8 * hh);  // [qc]: quadrant 2 h + qc
      const nlk_f4* const x0v = x0w;
      float S[NS][2][4];
#pragma unroll
      for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) S[a][q][j] = 0.f;
      float Ri[8], Rp[8];
      uint32_t oi, op, oin, opn;
      slot_offs(0, oi, op);
      const uint32_t oi0 = oi, op0 = op;
      row_read(oi, 0, Ri);
      if (HP) row_read(op, 0, Rp);
      slot_offs(1, oin, opn);
      for (int ch = 0; ch < CH; ++ch)
      for (int b = 0; b < nb; ++b) {
        const bool wrap = b + 1 == nb;
        const int chn = wrap ? min(ch + 1, CH - 1) : ch;
#ifdef NLK_G8_EXP_LDSROWS   // (timing experiment: the cost of staging two 18 x 18 windows per channel)
        if (HP && b == 0) {
          float* win = stash + nlk_g8_stash_floats<CH, SEP>();
          const uint32_t back = 10u * (uint32_t)g.w + 10u;
          const uint32_t wo = (o_first > back ? o_first - back : o_first) + (uint32_t)ch * (uint32_t)npix;
          float tmp[12];
#pragma unroll
          for (int kq = 0; kq < 6; ++kq) {
            const uint32_t idx = min((uint32_t)lane + 64u * kq, 323u), r = (idx * 3641u) >> 16, cx = idx - 18u * r;
            tmp[kq] = pbase[(size_t)(e_img + wo + r * (uint32_t)g.w + cx)];
            tmp[6 + kq] = pbase[(size_t)(e_prev + wo + r * (uint32_t)g.w + cx)];
          }
#pragma unroll
          for (int kq = 0; kq < 6; ++kq) {
            win[(lane + 64 * kq) % (NLK_G8_LDS_PAD / 8)] = tmp[kq];
            win[NLK_G8_LDS_PAD / 8 + (lane + 64 * kq) % (NLK_G8_LDS_PAD / 8)] = tmp[6 + kq];
          }
          nlk_wave_lds_order();
        }
#endif
        float Fi[2][4], Fp[2][4];
        fold_half(Ri, Fi);
        if (HP) fold_half(Rp, Fp);
        // the next batch's rows (after a channel's last batch the next channel's first; after the very last one a
        // harmless reload), requested now: a whole batch of transforms and statistics for them to arrive
        uint32_t o_in = wrap ? oi0 : oin, o_pn = wrap ? op0 : opn;
        if (HP)
          asm volatile("" : "+v"(Fi[0][0]), "+v"(Fi[0][1]), "+v"(Fi[0][2]), "+v"(Fi[0][3]), "+v"(Fi[1][0]), "+v"(Fi[1][1]),
                       "+v"(Fi[1][2]), "+v"(Fi[1][3]), "+v"(Fp[0][0]), "+v"(Fp[0][1]), "+v"(Fp[0][2]), "+v"(Fp[0][3]),
                       "+v"(Fp[1][0]), "+v"(Fp[1][1]), "+v"(Fp[1][2]), "+v"(Fp[1][3]), "+v"(o_in), "+v"(o_pn));
        else
          asm volatile("" : "+v"(Fi[0][0]), "+v"(Fi[0][1]), "+v"(Fi[0][2]), "+v"(Fi[0][3]), "+v"(Fi[1][0]), "+v"(Fi[1][1]),
                       "+v"(Fi[1][2]), "+v"(Fi[1][3]), "+v"(o_in));
        __builtin_amdgcn_sched_barrier(0);
        row_read(o_in, chn, Ri);
        if (HP) row_read(o_pn, chn, Rp);
        __builtin_amdgcn_sched_barrier(0);
        slot_offs(wrap ? 1 : b + 2, oin, opn);
        nlk_f4 Yi[2], Yp[2];
        if (b == 0) {
          Yi[0] = Yi[1] = nlk_f4{0.f, 0.f, 0.f, 0.f};
          fwd_half(Fi, Yi);
          if (hp == 0) { x0w[0] = -Yi[0]; x0w[1] = -Yi[1]; }
          nlk_wave_lds_order();  // (lanes reading what other lanes have just written)
          Yi[0] += x0v[0]; Yi[1] += x0v[1];
        } else {
          Yi[0] = x0v[0]; Yi[1] = x0v[1];
          fwd_half(Fi, Yi);
        }
        if (HP) {
          Yp[0] = x0v[0]; Yp[1] = x0v[1];
          fwd_half(Fp, Yp);
        }
        if constexpr (HP) {
          // the 8 candidates of the batch: valid previous patch / group membership, one bit per lane quad of a half
          const uint64_t vw = b < 8 ? vbits[0] : vbits[1], gw = b < 8 ? gbits[0] : gbits[1];
          const uint32_t vch = (uint32_t)(vw >> (8 * (b & 7))) & 0xffu, gch = (uint32_t)(gw >> (8 * (b & 7))) & 0xffu;
          if constexpr (MODE == 1) {
            const bool mixed = (vch & ~gch) != 0u;  // (wave-uniform) valid candidates that are not members
            const float mkc = (float)((~gch >> hp) & 1u);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float di = Yi[q][j], d = Yp[q][j];
                const float df = di - d;  // reference: :769-783
                if (b == 0) {  // (a channel's first batch assigns: these sums are not cleared per channel)
                  S[0][q][j] = d;
                  S[1][q][j] = d * d;
                  S[2][q][j] = df * df;
                } else {
                  S[0][q][j] += d;
                  S[1][q][j] = fmaf(d, d, S[1][q][j]);
                  S[2][q][j] = fmaf(df, df, S[2][q][j]);
                }
              }
            if (mixed) {
#pragma unroll
              for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) S[NS > 3 ? 3 : 0][q][j] = fmaf(mkc, Yp[q][j], S[NS > 3 ? 3 : 0][q][j]);
            }
          } else {
            const float mk = (float)((vch >> hp) & 1u);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float di = Yi[q][j], d = Yp[q][j];
                S[0][q][j] += di;
                S[1][q][j] = fmaf(di, di, S[1][q][j]);
                S[NS > 2 ? 2 : 0][q][j] += d;
                S[NS > 3 ? 3 : 0][q][j] = fmaf(d, d, S[NS > 3 ? 3 : 0][q][j]);
                const float df = di - d;  // smoother: :1659-1667
                S[NS > 4 ? 4 : 0][q][j] = fmaf(mk * df, df, S[NS > 4 ? 4 : 0][q][j]);
              }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              S[0][q][j] += Yi[q][j];
              S[1][q][j] = fmaf(Yi[q][j], Yi[q][j], S[1][q][j]);
            }
        }
        if (wrap) {
          // The channel is complete. Per statistic 8 registers (qc, a) in every lane, to be summed over the 8 candidates'
          // lanes (lane bits 3..5; bit 2 = the vertical parity, bits 0..1 = the horizontal index stay) - as a
          // reduce-scatter: v_permlane32_swap + add sends the two horizontal parities to the two halves of the
          // wavefront (8 -> 4 registers), v_permlane16_swap + add the vertical indices {0, 1} / {2, 3} to the even / odd
          // rows of 16 lanes (-> 2), a rotation by 8 lanes the last bit (-> 1): lane 32 qc + 8 a + 4 qr + b ends up with
          // the total of coefficient (2 a + qr, 2 b + qc).
          float T[NS];
          const bool hi8 = (lane & 8) != 0;
#pragma unroll
          for (int a = 0; a < NS; ++a) {
            float Z[4], W[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const auto z = __builtin_amdgcn_permlane32_swap(__float_as_uint(S[a][0][j]), __float_as_uint(S[a][1][j]), false, false);
              Z[j] = __uint_as_float(z[0]) + __uint_as_float(z[1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const auto z = __builtin_amdgcn_permlane16_swap(__float_as_uint(Z[j]), __float_as_uint(Z[j + 2]), false, false);
              W[j] = __uint_as_float(z[0]) + __uint_as_float(z[1]);
            }
            const float send = hi8 ? W[0] : W[1], keep = hi8 ? W[1] : W[0];
            T[a] = keep + nlk_dpp<NLK_DPP_ROR8>(send);
            // (the filter's three running sums are ASSIGNED by the next channel's first batch; the members' sum, which
            // only the batches with a non-member add to, and the other modes' sums start from zero)
            if (!(MODE == 1 && a < 3)) {
#pragma unroll
              for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) S[a][q][j] = 0.f;
            }
          }
          sep_gain(mode_tag, T, ch);
        }
      }
    };
    if (np0 == 0) pass_a(std::integral_constant<int, 0>{});
    else if (!SMO) pass_a(std::integral_constant<int, 1>{});
    else pass_a(std::integral_constant<int, 2>{});
    }

    nlk_wave_lds_order();  // (the gains were parked by other lanes)
    // ---------------- pass B: shrink, invert and aggregate the group members, 4 per step
    // (slot lo = 4*channel + member; slots without a member / channel read a valid patch
    // and their results are not used)
    // G16 (one channel, separable form): the weight plane needs no transform (its pixels are the constant 1) and there
    // is only one image plane, so all sixteen slots of a step carry the image patch of a member of their own - slot
    // (m, pl) = member n0 + 4 pl + m - and 20 members take 2 steps instead of 5. After the transposition register jm
    // of lane group pl is member n0 + 4 pl + jm; the tile is then updated in sixteen rounds, one member each: the
    // 16 lanes of group pl add the member's pixels to the image plane while the 16 lanes of the next group add its
    // weights to the weight plane (an LDS instruction may touch ONE member per plane: the same 8 instructions per
    // member as before, half the lanes idle).
    // M5 (three channels, separable form; round 6): FIVE members per step. The weight plane's lane group needs no
    // transform - its pixels are the constant 1 - so its four slots carry the three channels of a fifth member
    // (slot (g4 = c, plane 3) = member n0 + 4, channel c; the sixteenth slot idles): 20 members in 4 steps instead
    // of 5, a fifth of pass B's transforms, loads and shrinkage gone. After the transposition the fifth member's
    // pixels sit in the weight plane's lanes with the CHANNEL in the register index; three DPP row shifts per pixel
    // with a bank mask each (lane 4 c + si of a row takes lane 12 + si's register c) put them in front of their own
    // planes, the weight lanes keep the constant, and the member is aggregated by the same eight LDS instructions as
    // every other - no exec-masked rounds (round 5's version of the idea paid four of them per step and lost).
#ifndef NLK_G8_M5
#define NLK_G8_M5 1
#endif
    constexpr bool M5 = NLK_G8_M5 && SEPB && CH == 3;
    constexpr int MPS = G16 ? 16 : (M5 ? 5 : 4);  // members per step
    constexpr int NPX = M5 ? 5 : 4;               // members whose pixels a step leaves in registers
    const bool wlane = SEPB && spl == 3;          // (M5) a lane of the weight plane's group
    const int bchc = G16 ? 0 : (M5 ? (wlane ? min(g4, CH - 1) : spl) : min(bch, CH - 1));
    const int bmm = G16 ? 4 * spl + g4 : (M5 ? (wlane ? 4 : g4) : bm);
    auto member_off = [&](int n0) -> uint32_t {   // (element offset inside an image: channel plane + patch)
      const int n = min(n0 + bmm, nagg - 1);
      uint32_t qm;
      if constexpr (M5) {  // (a step of five may straddle the two list words, and a bpermute's operand is the SOURCE lane's)
        const uint32_t q0 = nlk_bperm_u(greg[0], n & 63), q1 = nlk_bperm_u(greg[1], n & 63);
        qm = n < 64 ? q0 : q1;
      } else {
        qm = nlk_bperm_u(n < 64 ? greg[0] : greg[1], n & 63);
      }
      return (uint32_t)bchc * (uint32_t)npix + (uint32_t)(nlk_y(qm) * g.w + nlk_x(qm));
    };
    // (smoother) the difference image previous - image laid out with the frame (k_layout): ONE row set per member
    // instead of two, no subtraction here - the same float operation on the same operands, done once per pixel.
    // Pass B of the smoother loads 8 x 16 bytes per lane and step, every lane from cache lines of its own: the
    // vector L1 (~2 lines a cycle) is what the separable form's shorter steps waited for (round 6: SMO1 group
    // 1.316 ms -> 1.097 with these loads compiled out; the Kronecker form 1.278 -> 1.236).
    // (Without a previous frame there is no difference image and every target is passed through: its coefficients
    // are set to zero below whatever was loaded - the image's own rows then.)
    const uint32_t e_bsrc = (SMO && tl.diff) ? (uint32_t)(tl.diff - pbase) : e_src;
    float R[16], F[4][4];
    nlk_rows_load32(pbase, e_bsrc + member_off(0), rowaB, rowbB, R);
    uint32_t offn = member_off(MPS);
    // Where every member lands in the tile, worked out once per target with one member per lane: its offset
    // (floats) inside a plane, and one bit per member "inside the tile" (entries past the last member count as
    // inside). A step whose four members are all inside - every step of a temporal target: the tile's halo is
    // the temporal radius - then aggregates in straight-line code. (Before, each member brought ~20 scalar
    // instructions and five branches with it: a sixth of the kernel's time, measured with the member loop
    // compiled out.)
    uint32_t mbase[2];
    uint64_t inside[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int lx = nlk_x(greg[m]) - rx0, ly = nlk_y(greg[m]) - ry0;
      const bool in = lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh;
      mbase[m] = (uint32_t)(ly * rwp + lx);
      inside[m] = __ballot(in || lane + 64 * m >= nagg);
    }
    // the reference adds the same per-coefficient terms once per group member
    float vp = nlk_wave_sum_dpp(part_sum) * (float)nagg;
    if (passthrough) vp = 0.f;
    const float wgt = 1.f / (vp > 1e-6f ? vp : 1e-6f);
    float ww[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ww[kk] = wgt * win[kk];
    const int bst = G16 ? 0 : (M5 ? bchc : min(bch, NSLOT - 1));  // stash slot of the lane's load slot: image channel, weights, or nothing
    const bool slot_is_channel = (G16 || M5) ? true : bch < CH;
    // gains of the lane's four coefficients of quadrant q at + 4 * q (separable) / + 16 * q (Kronecker)
    const float* st_g = stash + bst * SST + (SEPB ? 16 * si : 4 * g4);
    const float* st_m = st_g + 64;
    constexpr int QST = SEPB ? 4 : 16;
    // One step's transforms: the rows of members n0 .. n0 + MPS - 1 (in R: requested a step ahead) -> PX[m][kk], the
    // lane's four pixels (aggregation role above) of member n0 + m, plane aplane; requests the next step's rows.
    auto step_px = [&](int n0, float (&PX)[NPX][4]) {
      nlk_f4 Y[4];
      // The smoother's update (1 - a) A + a B of a member's coefficients (A image, B previous frame,
      // reference: :1775) is A + a (B - A), and every step from here to the frame is linear: the member's
      // pixels are its image patch + IDCT(a . DCT(previous patch - image patch)), and the image patches,
      // added with the members' weights, sum to image x weight plane. So only the SECOND term goes through
      // the transforms - one forward transform of the pixel difference (R holds it: the difference image) instead
      // of two - and the image term is added where the tile leaves for the frame (the flush below): image x the
      // tile's weight plane.
      nlk_fold(R, F);
#ifdef NLK_G8_EXP_NOROWSB   // (timing experiment: pass B's rows loaded once per target)
      asm volatile("" : "+v"(R[0]), "+v"(R[5]), "+v"(R[10]), "+v"(R[15]) : "v"(offn));
#else
      nlk_rows_load32(pbase, e_bsrc + offn, rowaB, rowbB, R);
#endif
#pragma unroll
      for (int q = 0; q < 4; ++q) Y[q] = nlk_f4{0.f, 0.f, 0.f, 0.f};
      offn = member_off(n0 + 2 * MPS);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SEPB) {
        nlk_sep_fwd4<PRIO>(F, sE, Y);
      } else {
        nlk_mfma_fwd<true, PRIO>(F, dA, Y);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const nlk_f4 gq = *reinterpret_cast<const nlk_f4*>(st_g + QST * q);
        const nlk_f4 mq = *reinterpret_cast<const nlk_f4*>(st_m + QST * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (smoother: weight / unused slots carry gain 0 and their value - the constant's DCT - in the mean half; a
          // pass-through target reads its own patch as "previous" - the difference is zero - and its gains
          // come from the Wiener formula with beta_x = 0, i.e. may be 0 / 0: not used)
          if (SMO) Y[q][j] = slot_is_channel ? (passthrough ? 0.f : gq[j] * Y[q][j]) : mq[j];
          else Y[q][j] = fmaf(gq[j], Y[q][j], mq[j]);
        }
      }
      // PX[m][kk]: the lane's four pixels (aggregation role above) of member n0 + m, plane aplane
      if constexpr (SEPB) {
        nlk_f4 X[4];
        nlk_sep_inv4<PRIO>(Y, sG, X);
        // unfold: rows si (O[0]) and 7 - si (O[1]) of the slot's patch, 8 columns each
        float O[2][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float e0 = X[0][c] + X[1][c], e1 = X[0][c] - X[1][c];
          const float o0 = X[2][c] + X[3][c], o1 = X[2][c] - X[3][c];
          O[0][c] = e0 + o0; O[0][7 - c] = e1 + o1; O[1][c] = e0 - o0; O[1][7 - c] = e1 - o1;
        }
        // One LDS instruction takes one register of all 64 lanes - sixteen patches at once, four of them in every
        // plane, which may overlap: the tile updates are plain read-modify-writes, so an instruction must touch ONE
        // member per plane. Transpose the step's member index (lane bits 4..5) with the column index modulo 4
        // (register) by the row-swap instructions - 4 per group of four registers, no adds: afterwards register
        // 4 h + m of a row holds column 4 h + g4 of member m (tools/ubench/permlane_swap.hip for what the swaps do).
#pragma unroll
        for (int sel = 0; sel < 2; ++sel)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float* o = &O[sel][4 * h];
            const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[0]), __float_as_uint(o[2]), false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[1]), __float_as_uint(o[3]), false, false);
            const auto u = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
            const auto v = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
            PX[0][2 * sel + h] = __uint_as_float(u[0]); PX[1][2 * sel + h] = __uint_as_float(u[1]);
            PX[2][2 * sel + h] = __uint_as_float(v[0]); PX[3][2 * sel + h] = __uint_as_float(v[1]);
          }
        if constexpr (M5) {
          // the fifth member: register c of the weight group's lanes -> the lanes of plane c (row_shl 12 / 8 / 4 inside
          // the rows of 16 lanes, written to bank c only); the weight group's own lanes keep the constant 1
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            int p4 = __float_as_int(1.f);
            p4 = __builtin_amdgcn_update_dpp(p4, __float_as_int(PX[0][kk]), 0x10C /* row_shl:12 */, 0xF, 0x1, false);
            p4 = __builtin_amdgcn_update_dpp(p4, __float_as_int(PX[1][kk]), 0x108 /* row_shl:8 */, 0xF, 0x2, false);
            p4 = __builtin_amdgcn_update_dpp(p4, __float_as_int(PX[2][kk]), 0x104 /* row_shl:4 */, 0xF, 0x4, false);
            PX[4][kk] = __int_as_float(p4);
          }
          // ... and the weight plane's pixels of the other four members: the constant
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) PX[m][kk] = wlane ? 1.f : PX[m][kk];
        }
      } else {
        nlk_f4 Z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) Z[q] = nlk_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float y[4] = {Y[q][0], Y[q][1], Y[q][2], Y[q][3]};
          Z[q] = nlk_mfma_q<false>(y, dI[q], Z[q]);
        }
        // register m of Z = member n0+m, plane g4, folded pixel lo
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float e0 = Z[0][m] + Z[1][m], e1 = Z[0][m] - Z[1][m];
          const float o0 = Z[2][m] + Z[3][m], o1 = Z[2][m] - Z[3][m];
          PX[m][0] = e0 + o0; PX[m][1] = e1 + o1; PX[m][2] = e0 - o0; PX[m][3] = e1 - o1;
        }
      }
    };
    // the tile update of a step whose four members all lie inside the tile (straight-line code)
    auto agg_fast = [&](int n0, const float (&PX)[NPX][4]) {
        // (read with every lane active: a lane read inside `if (agg_on)` is only defined for the lanes that are
        // on there, and the member index runs over all 64)
        int tile_off[NPX];
#pragma unroll
        for (int m = 0; m < NPX; ++m)   // (a step of five may straddle the two 64-entry words)
          tile_off[m] = (n0 + m) < 64 ? __builtin_amdgcn_readlane((int)mbase[0], (n0 + m) & 63)
                                      : __builtin_amdgcn_readlane((int)mbase[1], (n0 + m) & 63);
#pragma unroll
        for (int m = 0; m < NPX; ++m) {
          if (agg_on) {  // (CH = 3: every lane owns a plane)
            float* dst = smem + tile_off[m];
            float old[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) old[kk] = dst[poff[kk]];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dst[poff[kk]] = fmaf(ww[kk], PX[m][kk], old[kk]);
          }
        }
    };
    // Software pipeline over the steps (round 6): a step is one long dependent chain - rows, fold, two stages of
    // matrix products, shrink, two more stages, unfold, transposition, then four read-modify-writes of the tile one
    // behind the other (the members overlap: they cannot be reordered) - and pass B kept the FP32 datapath busy
    // only 59 % of its time against pass A's 80 %. When the first `npipe` members all lie inside the tile (every
    // temporal target: the tile's halo is the temporal radius) the tile update of step n - 1 stands in the same
    // basic block as the transforms of step n, so that the scheduler can put the LDS latencies of the one under
    // the matrix products of the other; two PX buffers swap roles, no copies.
    int n_start = 0;
#ifdef NLK_G8_PIPE   // (measured slower - 10 to 15 registers spilled at 168: C2 group 0.711 -> 0.742 ms, profiles/README.md round 6)
    if constexpr (!G16 && !M5) {
      const int npipe = min(nagg & ~3, 64);
      const uint64_t need = npipe >= 64 ? ~0ull : ((1ull << npipe) - 1ull);
      if (npipe >= 8 && (inside[0] & need) == need) {
        float PXa[4][4], PXb[4][4];
        step_px(0, PXa);
        int n0 = 4;
        for (; n0 + 4 < npipe; n0 += 8) {
          step_px(n0, PXb);
          agg_fast(n0 - 4, PXa);
          step_px(n0 + 4, PXa);
          agg_fast(n0, PXb);
        }
        if (n0 < npipe) {
          step_px(n0, PXb);
          agg_fast(n0 - 4, PXa);
          agg_fast(n0, PXb);
        } else {
          agg_fast(n0 - 4, PXa);
        }
        n_start = npipe;
      }
    }
#endif
    for (int n0 = n_start; n0 < nagg; n0 += MPS) {
      float PX[NPX][4];
      step_px(n0, PX);
      if constexpr (G16) {
#pragma unroll
        for (int jm = 0; jm < 4; ++jm)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int n = n0 + 4 * t + jm;   // (wave-uniform) the member this round adds
            if (n >= nagg) continue;
            const bool act_img = spl == t, act_w = spl == ((t + 1) & 3);
            const bool in = ((n < 64 ? inside[0] >> n : inside[1] >> (n - 64)) & 1ull) != 0;
            const int toff = n < 64 ? __builtin_amdgcn_readlane((int)mbase[0], n) : __builtin_amdgcn_readlane((int)mbase[1], n - 64);
            const uint32_t q = n < 64 ? __builtin_amdgcn_readlane(greg[0], n) : __builtin_amdgcn_readlane(greg[1], n - 64);
            if (act_img || act_w) {
              float px[4];
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) px[kk] = act_img ? PX[jm][kk] : 1.f;
              if (in) {
                float* dst = smem + toff + (act_w ? plane : 0);
                float old[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) old[kk] = dst[poff[kk]];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dst[poff[kk]] = fmaf(ww[kk], px[kk], old[kk]);
              } else {
                // a member outside the tile: straight to the frame, the smoother's image term with it
                const int qx = nlk_x(q), qy = nlk_y(q);
                int goff[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) goff[kk] = ((kk & 2) ? 7 - si : si) * g.w + 4 * (kk & 1) + g4;
                float* dst = acc + (act_w ? npix : (size_t)0) + (size_t)qy * g.w + qx;
                if (SMO && act_img) {
                  const float* ip = src + (size_t)qy * g.w + qx;
#pragma unroll
                  for (int kk = 0; kk < 4; ++kk) px[kk] += ip[goff[kk]];
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) unsafeAtomicAdd(dst + goff[kk], ww[kk] * px[kk]);
              }
            }
          }
        continue;
      }
      // "inside the tile" bits of the step's members (a step of five may straddle the two 64-entry words)
      uint64_t inb = n0 < 64 ? inside[0] >> n0 : inside[1] >> (n0 - 64);
      if (NPX > 4 && n0 < 64 && n0 + NPX > 64) inb |= inside[1] << (64 - n0);
      constexpr uint32_t in_all = (1u << NPX) - 1u;
      if (((uint32_t)inb & in_all) == in_all && n0 + NPX <= nagg) {
        agg_fast(n0, PX);
        continue;
      }
#pragma unroll
      for (int m = 0; m < NPX; ++m) {
        if (n0 + m >= nagg) break;
        const uint32_t q = (n0 + m) < 64 ? __builtin_amdgcn_readlane(greg[0], n0 + m)
                                         : __builtin_amdgcn_readlane(greg[1], n0 + m - 64);
        const int qx = nlk_x(q), qy = nlk_y(q);
        float px[4] = {PX[m][0], PX[m][1], PX[m][2], PX[m][3]};
        const int lx = qx - rx0, ly = qy - ry0;
        if (lx >= 0 && ly >= 0 && lx + PSZ <= rw && ly + PSZ <= rh) {
          if (agg_on) {
            float* dst = smem + ly * rwp + lx;
            float old[4];  // the four pixels are distinct: read them together
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) old[kk] = dst[poff[kk]];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dst[poff[kk]] = fmaf(ww[kk], px[kk], old[kk]);
          }
        } else if (agg_on) {
          // a member outside the tile (never with the smoother's own halo; kept complete): straight to the frame,
          // the smoother's image term with it
          // (the frame offsets of the lane's four pixels are worked out HERE - the rare path - not kept in registers
          // across the passes)
          int goff[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int r = SEPB ? ((kk & 2) ? 7 - si : si) : ((kk & 2) ? 7 - pi : pi);
            const int c = SEPB ? 4 * (kk & 1) + g4 : ((kk & 1) ? 7 - pj : pj);
            goff[kk] = r * g.w + c;
          }
          float* dst = acc + (size_t)aplane * npix + (size_t)qy * g.w + qx;
          if (SMO && aplane < CH) {
            const float* ip = src + (size_t)aplane * npix + (size_t)qy * g.w + qx;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) px[kk] += ip[goff[kk]];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) unsafeAtomicAdd(dst + goff[kk], ww[kk] * px[kk]);
        }
      }
    }
  }

  // ---------------- flush the tile accumulator
  __syncthreads();

  if (tl.slab) {
    // deterministic mode (k_gather.h): the planes as they stand, into this tile's slab
    if (lane == 0) {
      tl.tflag[tile_id] = any_target;
      if (any_target) { atomicAdd(&tl.tcount[0], 1); atomicAdd(&tl.tcount[1 + tile_y], 1); }
    }
    if (any_target) {
      if (SMO) {
        // (the smoother's tile holds weighted sums of member - image: the image term, image x weight plane, joins
        // here, so that a slab means what every other slab means)
        const float* wsp = smem + CH * plane;
        for (int p = 0; p < CH; ++p) {
          const float* ip = src + (size_t)p * npix + (size_t)ry0 * g.w + rx0;
          for (int y = 0; y < rh; ++y)
            for (int xx = lane; xx < rw; xx += 64) {
              const float wv = wsp[y * rwp + xx];
              if (wv != 0.f) smem[p * plane + y * rwp + xx] = fmaf(ip[(size_t)y * g.w + xx], wv, smem[p * plane + y * rwp + xx]);
            }
        }
        __syncthreads();
      }
      nlk_f4* dst = reinterpret_cast<nlk_f4*>(tl.slab + (size_t)tile_id * (CH + 1) * plane);
      for (int i = lane; i < (CH + 1) * plane / 4; i += 64) dst[i] = reinterpret_cast<const nlk_f4*>(smem)[i];
    }
  } else {
    // untouched entries skipped. A wavefront's flush is paced by the atomics it may have in flight (16-32 per
    // wave at ~1000-3000 cycles each under load: MI355X_MICROARCH.md, float atomic add), i.e. by the NUMBER of
    // atomic instructions: the region is walked as one run of rw x rh entries, 64 per instruction (9 per plane
    // of a 26 x 22 tile instead of the 11 that two 26-entry rows per instruction take), (x, y) kept per lane
    // by adding 64 = qstep rows + rstep columns with a carry.
    const int qstep = 64 / rw, rstep = 64 - qstep * rw;
    const float* wsp = smem + CH * plane;
    for (int p = 0; p <= CH; ++p) {
      const float* sp = smem + p * plane;
      float* dp = acc + (size_t)p * npix + (size_t)ry0 * g.w + rx0;
      const float* ip = src + (size_t)min(p, CH - 1) * npix + (size_t)ry0 * g.w + rx0;
      int y = lane / rw, xx = lane - y * rw;
#pragma unroll 3
      for (; y < rh; ) {
        float v = sp[y * rwp + xx];
        if (SMO && p < CH) {
          // the smoother's tile holds weighted sums of (member - image): the image term of every member that
          // landed on this pixel is image x the tile's weight, added here - the accumulator keeps its meaning
          // (weighted sums of member pixels) for whoever normalises or reduces it
          const float wv = wsp[y * rwp + xx];
          if (wv != 0.f) v = fmaf(ip[(size_t)y * g.w + xx], wv, v);
        }
        if (v != 0.f) unsafeAtomicAdd(dp + (size_t)y * g.w + xx, v);
        xx += rstep; y += qstep;
        if (xx >= rw) { xx -= rw; ++y; }
      }
    }
  }
}
