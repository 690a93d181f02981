// k_match.h — exhaustive block matching + ranked k-NN selection + group
// membership, for every target of the patch grid
// (reference: src/nlkalman.c:605-609, 630-707, 725-732, 779-793, 857, 931).
//
// One workgroup owns a tile of TGX x TGY grid targets and stages the part of
// the matching image those targets can reach (patches + search halo) in LDS,
// planar per channel, with coalesced row reads from HBM. Lane = candidate
// (lane + 64*m for the m-th round). The squared distance is accumulated in the
// reference's element order (hy, hx, c) with one rounding per subtract, multiply
// and add, so that the ranking is reproducible bit for bit by the CPU
// restatement. A wavefront takes a block of 4 x 2 targets: where all eight search
// the same full window of <= 448 candidates (the bulk of any frame: 121 in a
// temporal one, 441 in a spatial one) the
// squared differences of the pixels they share are computed once and added to
// each target's sum in that target's own order (nlk_match_block: identical sums,
// 2.1x fewer subtractions, multiplications and LDS reads); otherwise one target at
// a time (nlk_match_target). The LDS row stride is chosen so that the 64 candidate
// reads of a wavefront are bank-conflict free for the dominant window width.
//
// Selection = exact k smallest under the (distance, window index) order, which
// is the prefix of the reference's stable ascending sort: a 32-step bitwise
// radix select on the float bits finds the k-th distance in registers
// (ballot + popcount, no LDS), ties are cut in window order, and only the k
// survivors are ranked against each other to produce the sorted list.
#pragma once
#include "nlk_common.h"

#include <type_traits>
#ifndef NLK_BM_BLOCK7
#define NLK_BM_BLOCK7 1  // spatial windows (<= 448 candidates) in blocks too
#endif
#ifndef NLK_BM_THREADS
#define NLK_BM_THREADS 256
#endif
#define NLK_BM_WAVES (NLK_BM_THREADS / 64)

struct NlkTile {
  int tgx, tgy;       // targets per tile
  int bx;             // targets per block along x: 4, or 2 (k_bm_topk<.., 2>: 12 x 12 patches, 8 wavefronts on an 8 x 4 tile)
  int ntx, nty;       // tiles
  int rwp, rh_max;    // LDS region: padded row stride (floats) / rows
  int ksel_max;       // capacity of the per-wave survivor arrays
  int halo;           // search halo held in LDS (windows reaching further read HBM/L2)
  int block;          // 4 x 2-target blocks share their squared differences (0: NLK_MATCH_NOBLOCK, target by target)
  int threads;        // k_bm_topk: threads per workgroup (256, or 512 for tiles of 8 x 8 targets)
  int order;          // summation order of the distances: 0 = the reference's (exact), 1 = block-summed (opt-in; 8 x 8 patches)
};

// OR over the 64 lanes, in every lane. (Round 6: four DPP steps inside the rows of 16 lanes and the two row-swap
// instructions instead of six ds_bpermute round trips per half word - no LDS instruction, no LDS latency in the
// per-target epilogue; tools/ubench/permlane_swap.hip for what the swaps do.)
__device__ __forceinline__ uint32_t nlk_wave_or32(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140 /* row_mirror */, 0xF, 0xF, true);
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = a[0] | a[1];
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return b[0] | b[1];
}
__device__ __forceinline__ uint64_t nlk_wave_or(uint64_t v) {
  return (uint64_t)nlk_wave_or32((uint32_t)v) | ((uint64_t)nlk_wave_or32((uint32_t)(v >> 32)) << 32);
}

// Selection for one target whose window holds n <= 64*M candidates and whose sums of squared
// differences are in acc[] (candidate lane + 64*m in acc[m]). Leaves the k kept candidates, sorted, in
// sel[0..k).
// wxy[m] = (wy << 16) | wx of candidate lane + 64*m in its window: the same order as the window index
// wy * nwx + wx, and no division when the kept candidates are turned into coordinates.
template <int PSZ, int CH, int M>
__device__ __forceinline__ void nlk_match_select(const float (&acc)[M], const uint32_t (&wxy)[M], int n, int k,
                                                 int x0, int y0, uint64_t* __restrict__ surv,
                                                 uint32_t* __restrict__ sel, int lane) {
  uint32_t key[M];
  bool ok[M];
  const float norm = (float)(PSZ * PSZ * CH);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float q = acc[m] / norm;  // IEEE division (reference: src/nlkalman.c:701)
    key[m] = __float_as_uint(q > 0.f ? q : 0.f);
    ok[m] = lane + 64 * m < n;
  }

  // --- k-th smallest distance by bitwise radix select (keys are >= +0: uint order)
  // The set of candidates still matching the prefix is kept as one 64-bit lane
  // mask per round in scalar registers; per bit only the bit test is vector work.
  int kk = k;                    // how many of the still-undecided candidates are wanted
  uint64_t alive[M], less[M];    // undecided / already known to be among the k smallest
  int nalive = 0;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    alive[m] = __ballot(ok[m]);
    less[m] = 0;
    nalive += __popcll(alive[m]);
  }
  // walk the key bits from the top; stop as soon as exactly kk candidates are undecided
  // (they are then all kept) — typically after ~16 of the 32 bits. (Starting below the keys' common prefix - an AND and
  // an OR over the wavefront, ~10 rounds of 16 mostly scalar instructions saved per target - was tried twice: in round
  // 4 with shuffles, in round 6 with DPP / row-swap reductions: exact, and C2 match 0.2390 -> 0.2413 ms, first frame
  // 0.743 -> 0.751: the rounds above the prefix are scalar work the kernel has room for, the reductions are vector work.)
#pragma unroll 1
  for (int b = 31; b >= 0 && nalive != kk; --b) {
    uint64_t one[M];
    int cnt0 = 0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      one[m] = __ballot((key[m] >> b) & 1u);
      cnt0 += __popcll(alive[m] & ~one[m]);
    }
    if (kk > cnt0) {  // the k-th smallest has bit b set: everything with a 0 here is smaller
#pragma unroll
      for (int m = 0; m < M; ++m) {
        less[m] |= alive[m] & ~one[m];
        alive[m] &= one[m];
      }
      kk -= cnt0;
      nalive -= cnt0;
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) alive[m] &= ~one[m];
      nalive = cnt0;
    }
  }
  // undecided candidates now either number exactly kk (all kept) or share one key
  // value, of which the first kk in window order (ascending candidate index) are kept
  int ntie = 0, npos = 0;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const uint64_t be = alive[m];
    const int my_tie = ntie + __popcll(be & lt_mask);
    ntie += __popcll(be);
    const bool mine = (be >> lane) & 1ull;
    const bool keep = ((less[m] >> lane) & 1ull) || (mine && my_tie < kk);
    const uint64_t bk = __ballot(keep);
    if (keep) {  // survivors as one 64-bit sort key: distance bits above, window position below
      const int pos = npos + __popcll(bk & lt_mask);
      surv[pos] = ((uint64_t)key[m] << 32) | wxy[m];
    }
    npos += __popcll(bk);
  }
  nlk_wave_lds_fence();

  // --- rank the k survivors among themselves (one 64-bit compare per pair)
  if (k <= 32) {
    // two lanes per survivor: lane p counts the first half of the list, lane p + 32 the second
    const int p = lane & 31, half = lane >> 5, kh = (k + 1) >> 1;
    const uint64_t mine = p < k ? surv[p] : ~0ull;
    const int j0 = half ? kh : 0;
    const int nh = (half ? k : kh) - j0;  // how many survivors this lane counts, from j0 on
    const uint64_t* sp = surv + j0;
    int rank = 0;
    // (four survivors per round: their LDS reads are in flight together - one after the other every compare waited
    // for its own read - and the loop's bookkeeping is paid once per four. A round may read up to three entries past
    // the list: still inside the workgroup's LDS - the lists of the other wavefronts follow -, and they do not count.)
    for (int j = 0; j < kh; j += 4) {
      uint64_t sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] = sp[j + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) rank += (j + u < nh) && sv[u] < mine;
    }
    rank += __shfl_xor(rank, 32, 64);
    if (lane < k) {
      const uint32_t mi = (uint32_t)mine;
      sel[rank] = nlk_pack_xy(x0 + (int)(mi & 0xFFFFu), y0 + (int)(mi >> 16));
    }
  } else {
    for (int base = 0; base < k; base += 64) {
      const int p = base + lane;
      const uint64_t mine = p < k ? surv[p] : ~0ull;
      int rank = 0;
      for (int j = 0; j < k; j += 4) {
        uint64_t sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = surv[j + u];  // (past the list: see above)
#pragma unroll
        for (int u = 0; u < 4; ++u) rank += (j + u < k) && sv[u] < mine;
      }
      if (p < k) {
        const uint32_t mi = (uint32_t)mine;
        sel[rank] = nlk_pack_xy(x0 + (int)(mi & 0xFFFFu), y0 + (int)(mi >> 16));
      }
    }
  }
  nlk_wave_lds_fence();
}

// Distances + selection for one target whose window holds n <= 64*M candidates.
// Leaves the k kept candidates, sorted, in sel[0..k).
// (forced inline: as a real call the LDS tile pointer becomes a generic one, the candidate reads
// turn into FLAT loads and the spatial search runs 3x slower)
// ORD = 1: the opt-in block-summed order (NLK_MATCH_ORDER=block; nlk_match_block_sum below): the four quarter
// patches summed separately, then added - the same number whichever path computes it.
template <int PSZ, int CH, int M, int ORD = 0>
__device__ __forceinline__ void nlk_match_target(const float* __restrict__ tile, int plane, int rwp,
                                        const float* __restrict__ tgt, int tplane, int trw,
                                        int cbase, int nwx, int n, int k, int x0, int y0,
                                        uint64_t* __restrict__ surv, uint32_t* __restrict__ sel,
                                        int lane) {
  int cq[M];
  uint32_t wxy[M];
  float acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i = min(lane + 64 * m, n - 1);
    const int wy = i / nwx, wx = i - wy * nwx;
    cq[m] = cbase + wy * rwp + wx;
    wxy[m] = ((uint32_t)wy << 16) | (uint32_t)wx;
    acc[m] = 0.f;
  }
  if constexpr (ORD == 1) {
    constexpr int step = PSZ / 2;
    float sb[2][2][M];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < M; ++m) sb[i >> 1][i & 1][m] = 0.f;
#pragma unroll
    for (int sy = 0; sy < 2; ++sy) {
#pragma unroll 1
      for (int hy = sy * step; hy < (sy + 1) * step; ++hy) {
        const float* trow = tgt + hy * trw;
#pragma unroll
        for (int hx = 0; hx < PSZ; ++hx)
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float tv = trow[c * tplane + hx];
#pragma unroll
            for (int m = 0; m < M; ++m) {
              const float e = tile[c * plane + cq[m] + hy * rwp + hx] - tv;
              sb[sy][hx / step][m] = fmaf(e, e, sb[sy][hx / step][m]);
            }
          }
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = (sb[0][0][m] + sb[0][1][m]) + (sb[1][0][m] + sb[1][1][m]);
    nlk_match_select<PSZ, CH, M>(acc, wxy, n, k, x0, y0, surv, sel, lane);
    return;
  }
#pragma unroll 1
  for (int hy = 0; hy < PSZ; ++hy) {
#pragma clang fp contract(off)
    // the target patch is wave-uniform (broadcast LDS reads, or scalar loads on
    // the image path)
    const float* trow = tgt + hy * trw;
#pragma unroll
    for (int hx = 0; hx < PSZ; ++hx)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float tv = trow[c * tplane + hx];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float e = tile[c * plane + cq[m] + hy * rwp + hx] - tv;
          const float e2 = e * e;
          acc[m] = acc[m] + e2;
        }
      }
  }
  nlk_match_select<PSZ, CH, M>(acc, wxy, n, k, x0, y0, surv, sel, lane);
}

// Distances of a block of BX x BY grid-adjacent targets with the same full window (side 2 wsz + 1,
// n <= 64 M candidates, lane + 64 m = candidate), all inside the LDS region. Targets half a patch apart
// share three quarters of their pixels, and the squared difference of a pixel for a given candidate
// OFFSET is the same number whichever target it is summed for: it is computed once per pixel of the
// block's union and added to the accumulator of every target whose patch holds the pixel. Rows
// ascending, columns ascending inside a row, channels innermost: every target still receives its terms
// in the reference's (hy, hx, c) order with one rounding per subtract, multiply and add, i.e. the sums
// are bit-identical to nlk_match_target's; subtractions and multiplications drop by 2.1x (4 x 2 blocks).
// The targets' own pixels are the same for every lane: they come from the image through the SCALAR cache (timg =
// the union region's first pixel in plane 0, row stride w, plane stride npix) instead of as broadcast LDS reads -
// the LDS array, not the vector ALU, is what the row loops keep busiest (54 LDS instructions per row of a 2 x 2
// block, a third of their cycles for values that need no lane), and a vector instruction takes one scalar operand.
#ifndef NLK_BM_SCALAR_TARGET
#define NLK_BM_SCALAR_TARGET 1
#endif
template <int PSZ, int CH, int BX, int M, bool B0, bool B1>
__device__ __forceinline__ void nlk_block_rows(const float* __restrict__ tile, int plane, int rwp, int tbase,
                                               const int (&cq)[M], int ry0, int ry1, float (&acc)[2][BX][M],
                                               const float* __restrict__ timg, int w, size_t npix) {
  constexpr int step = PSZ / 2, UW = (BX - 1) * step + PSZ;
  const float* splane[CH];  // (one base per plane, set once; a row is a 32-bit offset from it)
#pragma unroll
  for (int c = 0; c < CH; ++c) splane[c] = timg + c * npix;
#pragma unroll 1
  for (int ry = ry0; ry < ry1; ++ry) {
#pragma clang fp contract(off)
    const float* trow = tile + tbase + ry * rwp;
    const int soff = ry * w;
    float tvs[CH][UW];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float* srow = splane[c] + soff;
#pragma unroll
      for (int rx = 0; rx < UW; ++rx) tvs[c][rx] = NLK_BM_SCALAR_TARGET ? srow[rx] : trow[c * plane + rx];
    }
#pragma unroll
    for (int rx = 0; rx < UW; ++rx)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float tv = tvs[c][rx];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float e = tile[c * plane + cq[m] + ry * rwp + rx] - tv;
          const float e2 = e * e;
#pragma unroll
          for (int bx = 0; bx < BX; ++bx)
            if (rx >= bx * step && rx < bx * step + PSZ) {
              if (B0) acc[0][bx][m] = acc[0][bx][m] + e2;
              if (B1) acc[1][bx][m] = acc[1][bx][m] + e2;
            }
        }
      }
  }
}
// window position (wy << 16) | wx of candidate lane + 64 m
template <int M>
__device__ __forceinline__ void nlk_window_xy(int wsz, int n, int lane, uint32_t (&wxy)[M]) {
  const int nwx = 2 * wsz + 1;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i = min(lane + 64 * m, n - 1);
    const int wy = i / nwx, wx = i - wy * nwx;
    wxy[m] = ((uint32_t)wy << 16) | (uint32_t)wx;
  }
}
template <int PSZ, int CH, int BX, int M>
__device__ __forceinline__ void nlk_match_block(const float* __restrict__ tile, int plane, int rwp, int tbase,
                                                int wsz, int n, int lane, float (&acc)[2][BX][M],
                                                const float* __restrict__ timg, int w, size_t npix) {
  constexpr int step = PSZ / 2;
  const int nwx = 2 * wsz + 1;
  int cq[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i = min(lane + 64 * m, n - 1);
    const int wy = i / nwx, wx = i - wy * nwx;
    cq[m] = tbase + (wy - wsz) * rwp + (wx - wsz);
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
      for (int bx = 0; bx < BX; ++bx) acc[by][bx][m] = 0.f;
  }
  // (rows of the upper targets only, of both, of the lower targets only)
  nlk_block_rows<PSZ, CH, BX, M, true, false>(tile, plane, rwp, tbase, cq, 0, step, acc, timg, w, npix);
  nlk_block_rows<PSZ, CH, BX, M, true, true>(tile, plane, rwp, tbase, cq, step, PSZ, acc, timg, w, npix);
  nlk_block_rows<PSZ, CH, BX, M, false, true>(tile, plane, rwp, tbase, cq, PSZ, PSZ + step, acc, timg, w, npix);
}

// ---- The opt-in BLOCK-SUMMED order (round 6; NLK_MATCH_ORDER=block, never the default).
// The grid step is half a patch, so a patch is 2 x 2 quarter patches of step x step pixels ALIGNED with the grid, and
// a quarter patch belongs to the four targets around it. Its sum of squared differences for a candidate offset is
// computed ONCE - rows ascending, columns ascending, channels innermost, one fused multiply-add per term - and a
// target's distance is (s00 + s01) + (s10 + s11) of its four quarters: the additions are shared too, not only the
// subtractions and multiplications (per target and candidate of a 2 x 2 block: 9/4 quarters x 48 terms x {sub, fma}
// = 216 vector operations + 3 adds against 408 in the exact order). This is NOT the reference's (hy, hx, c) order:
// the distances differ from the exact mode's in their last bits (the reference binary itself is built -ffast-math
// and promises no order, CMakeLists.txt:10), the k-NN lists where two candidates are within those bits
// (tests/test_gpu_parity.py::test_block_summed_match_order). Every path - blocks, single targets, k_bm_wide -
// uses the same quarters, so the result does not depend on tiles, strips or bands.
template <int PSZ, int CH, int BX, int M>
__device__ __forceinline__ void nlk_match_block_sum(const float* __restrict__ tile, int plane, int rwp, int tbase,
                                                    int wsz, int n, int lane, float (&acc)[2][BX][M],
                                                    const float* __restrict__ timg, int w, size_t npix) {
  constexpr int step = PSZ / 2, NSX = BX + 1, UW = NSX * step;
  const int nwx = 2 * wsz + 1;
  int cq[M];
  float sb[3][NSX][M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i = min(lane + 64 * m, n - 1);
    const int wy = i / nwx, wx = i - wy * nwx;
    cq[m] = tbase + (wy - wsz) * rwp + (wx - wsz);
#pragma unroll
    for (int sy = 0; sy < 3; ++sy)
#pragma unroll
      for (int sx = 0; sx < NSX; ++sx) sb[sy][sx][m] = 0.f;
  }
  const float* splane[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) splane[c] = timg + c * npix;
#pragma unroll
  for (int sy = 0; sy < 3; ++sy) {
#pragma unroll 1
    for (int ry = sy * step; ry < (sy + 1) * step; ++ry) {
      const int soff = ry * w;
      float tvs[CH][UW];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float* srow = splane[c] + soff;
#pragma unroll
        for (int rx = 0; rx < UW; ++rx) tvs[c][rx] = srow[rx];
      }
#pragma unroll
      for (int rx = 0; rx < UW; ++rx)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float tv = tvs[c][rx];
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const float e = tile[c * plane + cq[m] + ry * rwp + rx] - tv;
            sb[sy][rx / step][m] = fmaf(e, e, sb[sy][rx / step][m]);
          }
        }
    }
  }
#pragma unroll
  for (int by = 0; by < 2; ++by)
#pragma unroll
    for (int bx = 0; bx < BX; ++bx)
#pragma unroll
      for (int m = 0; m < M; ++m)
        acc[by][bx][m] = (sb[by][bx][m] + sb[by][bx + 1][m]) + (sb[by + 1][bx][m] + sb[by + 1][bx + 1][m]);
}

// Group membership, records and mark word of one target whose sorted k-NN list is in sel[0..k)
// (reference: src/nlkalman.c:725-732, 779-793, 857, 931; smoother :1669-1676, :1844).
// STEP = grid step when known at compile time (no integer divisions), 0 = g.step.
template <int STEP = 0>
__device__ __forceinline__ void nlk_match_epilogue(const NlkGeom& g, size_t t, int px, int py, int prev_p,
                                          int k, const uint32_t* __restrict__ sel,
                                          uint32_t* __restrict__ grp,
                                          const uint8_t* __restrict__ vmap,
                                          uint32_t* __restrict__ topk, NlkTarget* __restrict__ tinfo,
                                          uint32_t* __restrict__ gcoords,
                                          uint64_t* __restrict__ marks, int lane) {
  const int step = STEP ? STEP : g.step;
  NlkTarget info = {0, 0, 0, prev_p, {0ull, 0ull}};
  // --- group membership: the first ntagg kept candidates that have a valid
  // previous patch, or (none valid) the first ntagg kept candidates
  int np0 = 0;
  uint64_t vb[2] = {0ull, 0ull};
  for (int base = 0; base < k; base += 64) {
    const int i = base + lane;
    uint32_t q = 0;
    int v = 0;
    if (i < k) {
      q = sel[i];
      topk[t * g.kmax + i] = q;
      v = prev_p ? vmap[(size_t)nlk_y(q) * g.w + nlk_x(q)] : 0;
    }
    const uint64_t b = __ballot(v);
    if (base < 128) vb[base >> 6] = b;
    const int slot = np0 + __popcll(b & ((1ull << lane) - 1ull));
    if (v && slot < g.ntagg) grp[slot] = q;
    np0 += __popcll(b);
  }
  nlk_wave_lds_fence();

  int nagg, mark;
  if (g.smoother) {
    nagg = min(np0, g.ntagg);
    mark = np0 > 0;  // reference: :1844
  } else {
    nagg = min(np0 ? np0 : k, g.ntagg);
    mark = !(g.have_prev && np0 == 0);  // reference: :931
  }
  uint64_t mbits = 0;
  const int side = 2 * g.R + 1;
  for (int base = 0; base < nagg; base += 64) {
    const int i = base + lane;
    if (i < nagg) {
      const uint32_t q = np0 ? grp[i] : sel[i];
      gcoords[t * g.gstride + i] = q;
      const int dx = nlk_x(q) - px, dy = nlk_y(q) - py;
      if (mark && g.R <= 3 && dx % step == 0 && dy % step == 0) {  // (R > 3: k_mask_commit_lists reads the lists)
        const int di = dx / step, dj = dy / step;
        mbits |= 1ull << ((dj + g.R) * side + di + g.R);
      }
    }
  }
  if (g.smoother && np0 == 0) {  // pass-through of the target patch
    nagg = 1;
    if (lane == 0) gcoords[t * g.gstride] = nlk_pack_xy(px, py);
  }
  mbits = nlk_wave_or(mbits);
  if (lane == 0) {
    info.nsel = k;
    info.np0 = np0;
    info.nagg = nagg;
    info.flags = prev_p | (mark << 1);
    info.vbits[0] = vb[0]; info.vbits[1] = vb[1];
    tinfo[t] = info;
    marks[t] = mbits;
  }
}

#ifndef NLK_BM7_WAVES
#define NLK_BM7_WAVES 4
#endif
// ORD: 0 = the reference's (hy, hx, c) summation order (bit-identical distances), 1 = block-summed (opt-in, above)
template <int PSZ, int CH, int MAXM, int BX = 4, int ORD = 0>  // BX x 2 targets per block (2: twice the wavefronts on the same tile)
__global__ void __launch_bounds__(512, MAXM == 7 ? (BX == 2 ? NLK_BM7_WAVES : 3) : ((MAXM == 2 && PSZ == 8) ? 8 : 2))  // (7 rounds in blocks: 168 registers, the LDS tile allows 3 wavefronts per SIMD)
k_bm_topk(const float* __restrict__ img, const uint8_t* __restrict__ vmap, NlkGeom g,
          NlkTile tl, uint32_t* __restrict__ topk, NlkTarget* __restrict__ tinfo,
          uint32_t* __restrict__ gcoords, uint64_t* __restrict__ marks,
          uint32_t* __restrict__ wide_list, uint32_t* __restrict__ wide_count) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: makes per-target addresses uniform
  const int nwaves = (int)(blockDim.x >> 6);  // 4, or 8 for tiles of 8 x 8 targets (NlkTile::threads)
  const int tile_id = nlk_xcd_tile(blockIdx.x, tl.ntx * tl.nty);
  if (tile_id >= tl.ntx * tl.nty) return;
  // Tiles on the image border are the slow ones: the blocks whose windows are clipped go target by target
  // (~2x a regular block). Dealt out in raster order the bottom tile row ran LAST and a launch ended on 60 slow
  // workgroups. So the bottom row goes first (then the top row), and inside every tile row the two border columns -
  // whatever an XCD's band ends on is a regular tile (C2 match 0.258 -> 0.2566 ms: the tail was short).
  const int ord_y = tile_id / tl.ntx, ord_x = tile_id - ord_y * tl.ntx;
  const int tile_y = ord_y == 0 ? tl.nty - 1 : ord_y - 1;
  const int tile_x = ord_x == 0 ? 0 : (ord_x == 1 ? tl.ntx - 1 : ord_x - 1);
  const int gx0 = tile_x * tl.tgx, gy0 = tile_y * tl.tgy;
  const int cx = min(tl.tgx, g.ngx - gx0), cy = min(tl.tgy, g.ngy - gy0);
  const int wmax = tl.halo;
  constexpr int step = PSZ / 2;

  // image region staged in LDS: the tile's patches + the halo of the dominant
  // window. (In a temporal frame the few targets without a valid previous patch
  // search a wider window, reference: src/nlkalman.c:637; they are queued
  // for k_bm_wide.)
  const int rx0 = max(gx0 * step - wmax, 0);
  const int rx1 = min((gx0 + cx - 1) * step + wmax + PSZ, g.w);
  const int ry0 = max(g.oy + gy0 * step - wmax, 0);
  const int ry1 = min(g.oy + (gy0 + cy - 1) * step + wmax + PSZ, g.h);
  const int rw = rx1 - rx0, rh = ry1 - ry0;
  const int rwp = tl.rwp;
  const int plane = rwp * tl.rh_max;

  float* tile = smem;                                  // [CH][rh_max][rwp]
  // (the tile size is kept even so that the 64-bit survivor array is 8-byte aligned)
  uint64_t* surv_all = (uint64_t*)(tile + ((CH * plane + 1) & ~1));  // [waves][ksel_max]
  uint32_t* sel_all = (uint32_t*)(surv_all + nwaves * tl.ksel_max);
  uint32_t* grp_all = sel_all + nwaves * tl.ksel_max;  // [waves][gstride]

  const size_t npix = (size_t)g.w * g.h;
  // stage the region: 8 rows per wavefront in flight (loads first, then the LDS
  // stores), so the HBM/L2 latency is paid once per batch, not once per row
  {
    const int nrows = CH * rh;  // rows of all channel planes; region width <= 128
    for (int r0 = wave; r0 < nrows; r0 += nwaves * 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = min(r0 + nwaves * j, nrows - 1);
        int c = 0, y = r;  // (r / rh by comparison: an integer division is ~35 scalar instructions, 16 per wavefront here)
#pragma unroll
        for (int cc = 1; cc < CH; ++cc)
          if (r >= cc * rh) { c = cc; y = r - cc * rh; }
        const float* src = img + c * npix + (size_t)(ry0 + y) * g.w + rx0;
        v0[j] = lane < rw ? src[lane] : 0.f;
        v1[j] = lane + 64 < rw ? src[lane + 64] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + nwaves * j;
        if (r < nrows) {
          int c = 0, y = r;
#pragma unroll
          for (int cc = 1; cc < CH; ++cc)
            if (r >= cc * rh) { c = cc; y = r - cc * rh; }
          float* dst = tile + c * plane + y * rwp;
          if (lane < rw) dst[lane] = v0[j];
          if (lane + 64 < rw) dst[lane + 64] = v1[j];
        }
      }
    }
  }
  __syncthreads();

  // prev_p of the tile's targets: lane tt holds target tt's flag
  int rec_prev = 0;
  if (g.have_prev && lane < cx * cy) {
    const int ty = lane / cx, tx = lane - ty * cx;
    rec_prev = vmap[(size_t)(g.oy + (gy0 + ty) * step) * g.w + (gx0 + tx) * step];
  }
  uint64_t* surv = surv_all + wave * tl.ksel_max;
  uint32_t* sel = sel_all + wave * tl.ksel_max;
  uint32_t* grp = grp_all + wave * g.gstride;

  // one target: distances over its own (possibly clipped) window
  auto do_target = [&](int tt) {
    const int ty = tt / cx, tx = tt - ty * cx;
    const int gx = gx0 + tx, gy = gy0 + ty;
    const int px = gx * step, py = g.oy + gy * step;
    const size_t t = (size_t)gy * g.ngx + gx;
    const int prev_p = __builtin_amdgcn_readlane(rec_prev, tt);
    int k = prev_p ? g.npt : g.npx;
    NlkTarget info = {0, 0, 0, prev_p, {0ull, 0ull}};
    if (k <= 1) {
      // single-patch mode aggregates nothing in the filter (reference: :815-857);
      // the smoother passes the target patch through (reference: :1795-1804)
      if (g.smoother) {
        info.nagg = 1;
        if (lane == 0) gcoords[t * g.gstride] = nlk_pack_xy(px, py);
      }
      if (lane == 0) { tinfo[t] = info; marks[t] = 0; }
      return;
    }
    const int wsz = (g.smoother || prev_p) ? g.wsz_t : g.wsz_x;
    const int x0 = max(px - wsz, 0), x1 = min(px + wsz, g.w - PSZ) + 1;
    const int y0 = max(py - wsz, 0), y1 = min(py + wsz, g.h - PSZ) + 1;
    const int nwx = x1 - x0, n = nwx * (y1 - y0);
    k = min(k, n);
    // (MAXM of this kernel is sized for the dominant window: a clipped wide window near the
    // image border can lie inside the region and still have too many candidates)
    const bool in_lds = x0 >= rx0 && x1 - 1 + PSZ <= rx1 && y0 >= ry0 && y1 - 1 + PSZ <= ry1 &&
                        n <= 64 * MAXM;
    if (in_lds) {
      const int cbase = (y0 - ry0) * rwp + (x0 - rx0);
      // target patch from the LDS tile too (scalar loads from the image measured 6 % slower)
      const float* tl_tgt = tile + (py - ry0) * rwp + (px - rx0);
      if (n <= 128)
        nlk_match_target<PSZ, CH, 2, ORD>(tile, plane, rwp, tl_tgt, plane, rwp, cbase, nwx, n, k, x0,
                                          y0, surv, sel, lane);
      else if (MAXM <= 7 || n <= 448)
        nlk_match_target<PSZ, CH, (MAXM < 7 ? MAXM : 7), ORD>(tile, plane, rwp, tl_tgt, plane, rwp,
                                                              cbase, nwx, n, k, x0, y0, surv,
                                                              sel, lane);
      else
        nlk_match_target<PSZ, CH, MAXM, ORD>(tile, plane, rwp, tl_tgt, plane, rwp, cbase, nwx, n, k,
                                             x0, y0, surv, sel, lane);
    } else {
      // window leaves the LDS region (a target without a valid previous patch in a
      // temporal frame): queued for k_bm_wide, which stages a window of its own
      if (lane == 0) wide_list[atomicAdd(wide_count, 1u)] = (uint32_t)t;
      return;
    }
    nlk_match_epilogue<PSZ / 2>(g, t, px, py, prev_p, k, sel, grp, vmap, topk, tinfo, gcoords, marks, lane);
    __builtin_amdgcn_wave_barrier();
  };

  // A wavefront takes blocks of BX x BY targets. A block whose targets all search the same full
  // window of at most 448 candidates inside the LDS region (the bulk of a frame) shares the squared
  // differences between its targets (nlk_match_block, two or seven rounds of 64 candidates); any other
  // block - image border, a target without a valid previous patch in a temporal frame, windows of more
  // than 448 candidates - goes target by target.
  if (!tl.block) {  // (comparison variant, and small grids: the targets dealt out to the wavefronts one by one)
    for (int tt = wave; tt < cx * cy; tt += nwaves) do_target(tt);
    return;
  }
  constexpr int BY = 2;
  const int nbx = (tl.tgx + BX - 1) / BX, nby = (tl.tgy + BY - 1) / BY;
  for (int blk = wave; blk < nbx * nby; blk += nwaves) {
    const int bty = blk / nbx, tx0 = (blk - bty * nbx) * BX, ty0 = bty * BY;
    if (tx0 >= cx || ty0 >= cy) continue;
    bool regular = tx0 + BX <= cx && ty0 + BY <= cy && g.npt > 1 && g.npx > 1;
    int nprev = 0;
    if (regular)
      for (int j = 0; j < BX * BY; ++j)
        nprev += __builtin_amdgcn_readlane(rec_prev, (ty0 + j / BX) * cx + tx0 + j % BX) ? 1 : 0;
    const bool all_t = g.smoother || nprev == BX * BY;
    regular = regular && (all_t || nprev == 0);
    const int wsz = all_t ? g.wsz_t : g.wsz_x;
    const int nwx = 2 * wsz + 1, n = nwx * nwx;
    const int px0 = (gx0 + tx0) * step, py0 = g.oy + (gy0 + ty0) * step;
    constexpr int MB = (MAXM >= 7 && NLK_BM_BLOCK7) ? 7 : 2;  // rounds of 64 candidates a block may take
    regular = regular && n <= 64 * MB && wsz <= wmax && px0 - wsz >= rx0 && py0 - wsz >= ry0 &&
              px0 + (BX - 1) * step + wsz + PSZ <= min(rx1, g.w) && py0 + (BY - 1) * step + wsz + PSZ <= min(ry1, g.h);
    if (!regular) {
      for (int j = 0; j < BX * BY; ++j) {
        const int tx = tx0 + j % BX, ty = ty0 + j / BX;
        if (tx < cx && ty < cy) do_target(ty * cx + tx);
      }
      continue;
    }
    auto run_block = [&](auto mtag) {
      constexpr int M = decltype(mtag)::value;
      float acc[2][BX][M];
      if constexpr (ORD == 1)
        nlk_match_block_sum<PSZ, CH, BX, M>(tile, plane, rwp, (py0 - ry0) * rwp + (px0 - rx0), wsz, n, lane, acc,
                                            img + (size_t)py0 * g.w + px0, g.w, (size_t)g.w * g.h);
      else
        nlk_match_block<PSZ, CH, BX, M>(tile, plane, rwp, (py0 - ry0) * rwp + (px0 - rx0), wsz, n, lane, acc,
                                        img + (size_t)py0 * g.w + px0, g.w, (size_t)g.w * g.h);
      // (the window positions are only needed by the selection: worked out AFTER the row loops - carried through
      // them they were spilled to scratch at the 64-register budget, 85 MB of traffic per 1080p launch in round 3;
      // the asm keeps the compiler from hoisting the division back in front of the loops)
      int wsz_late = wsz;
      asm volatile("" : "+v"(wsz_late) : "v"(acc[0][0][0]), "v"(acc[1][BX - 1][M - 1]));
      uint32_t wxy[M];
      nlk_window_xy<M>(wsz_late, n, lane, wxy);
#pragma unroll 1
      for (int j = 0; j < BX * BY; ++j) {
        float a2[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a2[m] = 0.f;
#pragma unroll
        for (int by = 0; by < BY; ++by)
#pragma unroll
          for (int bx = 0; bx < BX; ++bx)
            if (j == by * BX + bx) {
#pragma unroll
              for (int m = 0; m < M; ++m) a2[m] = acc[by][bx][m];
            }
        const int tx = tx0 + j % BX, ty = ty0 + j / BX;
        const int gx = gx0 + tx, gy = gy0 + ty;
        const int px = gx * step, py = g.oy + gy * step;
        const size_t t = (size_t)gy * g.ngx + gx;
        const int prev_p = __builtin_amdgcn_readlane(rec_prev, ty * cx + tx);
        const int k = min(prev_p ? g.npt : g.npx, n);
        nlk_match_select<PSZ, CH, M>(a2, wxy, n, k, px - wsz, py - wsz, surv, sel, lane);
        nlk_match_epilogue<PSZ / 2>(g, t, px, py, prev_p, k, sel, grp, vmap, topk, tinfo, gcoords, marks, lane);
        __builtin_amdgcn_wave_barrier();
      }
    };
    if (MB == 2 || n <= 128) run_block(std::integral_constant<int, 2>{});
    else run_block(std::integral_constant<int, MB>{});
  }
}

// Targets whose window does not fit the tile of k_bm_topk (queued there): one
// wavefront per target stages the target's own window (patch + 2*wsz) in a
// private LDS region and runs the same distance / selection / epilogue code.
template <int PSZ, int CH, int MAXM, int ORD = 0>
__global__ void __launch_bounds__(NLK_BM_THREADS)
k_bm_wide(const float* __restrict__ img, const uint8_t* __restrict__ vmap, NlkGeom g, NlkTile tl,
          uint32_t* __restrict__ topk, NlkTarget* __restrict__ tinfo,
          uint32_t* __restrict__ gcoords, uint64_t* __restrict__ marks,
          uint32_t* __restrict__ wide_list, uint32_t* __restrict__ wide_count) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int step = PSZ / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rwp = tl.rwp, plane = rwp * tl.rh_max;  // per-wavefront region [CH][rh_max][rwp]
  const int per_wave = ((CH * plane + 1) & ~1) + ((3 * tl.ksel_max + g.gstride + 1) & ~1);
  float* tile = smem + (size_t)wave * per_wave;
  uint64_t* surv = (uint64_t*)(tile + ((CH * plane + 1) & ~1));
  uint32_t* sel = (uint32_t*)(surv + tl.ksel_max);
  uint32_t* grp = sel + tl.ksel_max;
  const size_t npix = (size_t)g.w * g.h;
  const int count = (int)*wide_count;
  for (int e = blockIdx.x * NLK_BM_WAVES + wave; e < count; e += gridDim.x * NLK_BM_WAVES) {
    const size_t t = wide_list[e];
    const int gy = (int)(t / g.ngx), gx = (int)(t - (size_t)gy * g.ngx);
    const int px = gx * step, py = g.oy + gy * step;
    const int prev_p = g.have_prev ? vmap[(size_t)py * g.w + px] : 0;
    int k = prev_p ? g.npt : g.npx;
    const int wsz = (g.smoother || prev_p) ? g.wsz_t : g.wsz_x;
    const int x0 = max(px - wsz, 0), x1 = min(px + wsz, g.w - PSZ) + 1;
    const int y0 = max(py - wsz, 0), y1 = min(py + wsz, g.h - PSZ) + 1;
    const int nwx = x1 - x0, n = nwx * (y1 - y0);
    k = min(k, n);
    // stage rows [y0, y1 + PSZ - 1) x [x0, x1 + PSZ - 1) of every channel
    const int rw = x1 - 1 + PSZ - x0, rh = y1 - 1 + PSZ - y0;
    for (int r0 = 0; r0 < CH * rh; r0 += 8) {  // 8 rows in flight, like k_bm_topk
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = min(r0 + j, CH * rh - 1);
        const int c = r / rh, y = r - c * rh;
        v[j] = lane < rw ? img[c * npix + (size_t)(y0 + y) * g.w + x0 + lane] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + j;
        if (r < CH * rh && lane < rw) {
          const int c = r / rh, y = r - c * rh;
          tile[c * plane + y * rwp + lane] = v[j];
        }
      }
    }
    nlk_wave_lds_fence();
    const float* tl_tgt = tile + (py - y0) * rwp + (px - x0);
    if (n <= 128)
      nlk_match_target<PSZ, CH, 2, ORD>(tile, plane, rwp, tl_tgt, plane, rwp, 0, nwx, n, k, x0, y0, surv,
                                        sel, lane);
    else if (MAXM <= 7 || n <= 448)
      nlk_match_target<PSZ, CH, (MAXM < 7 ? MAXM : 7), ORD>(tile, plane, rwp, tl_tgt, plane, rwp, 0, nwx,
                                                            n, k, x0, y0, surv, sel, lane);
    else
      nlk_match_target<PSZ, CH, MAXM, ORD>(tile, plane, rwp, tl_tgt, plane, rwp, 0, nwx, n, k, x0, y0,
                                           surv, sel, lane);
    nlk_match_epilogue<PSZ / 2>(g, t, px, py, prev_p, k, sel, grp, vmap, topk, tinfo, gcoords, marks, lane);
    __builtin_amdgcn_wave_barrier();
  }
}


