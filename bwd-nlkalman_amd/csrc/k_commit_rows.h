// k_commit_rows.h — the reach-1 row replay of the processed mask as a device function: the stand-alone kernel
// k_mask_commit_rows1 (k_commit.h) and workgroup 0 of k_group8m (k_group8m.h, NlkGTile::chase) both run it.
// See k_commit.h for the formulation (reference: src/nlkalman.c:597-600 skip test, :930-931 mark).
#pragma once
#include "nlk_common.h"

#define NLK_CR_BATCH 16  // rows per batch: the next batch's planes load while this one is replayed
// Rows [first, first + nrows) of the grid; the only state a row hands to the next - `a`, the columns marked
// from above - is kept per row in `astate` (row j's input at astate[j]), so that a grid replayed in bands
// (one call per band, in order) continues where the band before stopped.
// TAG: the decisions leave as words the consumers can poll while the replay is still running (k_group8m's
// workgroup 0 runs it, the others wait for the rows they need): tagged[j * 64 + word] = generation << 32 | bits,
// one indivisible 64-bit store at agent scope; a word counts once it carries the launch's generation (the buffer
// is cleared when it is allocated, generations start at 1 and only grow). Whole grid only (no row states).
template <int PF, bool TAG>
__device__ __forceinline__ void nlk_commit_rows1(const uint32_t* __restrict__ planes, uint32_t* __restrict__ actbits,
                                                 uint32_t* __restrict__ astate, uint64_t* __restrict__ tagged,
                                                 uint32_t generation, int ngx, int first, int nrows, int lane) {
  // columns of this word that exist
  const int nb = ngx - 32 * lane;
  const uint32_t colmask = nb <= 0 ? 0u : (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u));
  auto from_prev = [](uint32_t v) {  // lane l <- lane l-1 (lane 0 <- 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
  };
  auto from_next = [](uint32_t v) {  // lane l <- lane l+1 (lane 63 <- 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
  };
  const uint32_t* pp = planes + lane;
  uint32_t* ap = TAG ? nullptr : actbits + lane;
  uint32_t* sp = TAG ? nullptr : astate + lane;
  uint64_t* tp = TAG ? tagged + lane : nullptr;
  uint32_t a = (!TAG && first) ? sp[(size_t)first * 64] : 0u;  // marked from above
  auto load_batch = [&](uint32_t (&D)[PF][4], int jb) {
    const uint32_t* q = pp + (size_t)jb * 256;
#pragma unroll
    for (int r = 0; r < PF; ++r)
#pragma unroll
      for (int p = 0; p < 4; ++p) D[r][p] = q[(r * 4 + p) * 64];
  };
  auto run_batch = [&](const uint32_t (&D)[PF][4], int jb) {
#pragma unroll
    for (int r = 0; r < PF; ++r) {  // (rows past the band in its last batch: decisions and states land in rows the
                                    //  next band rewrites, or in the padding)
      const uint32_t g = D[r][0] & ~a;
      // carries, assuming no carry enters the word
      const uint32_t sw = g & ~(g << 1);
      const uint32_t er = g & ~(g + (sw & 0x55555555u));  // runs that start on an even bit
      const uint32_t c0 = g & ((er & 0x55555555u) | (~er & 0xAAAAAAAAu));
      uint32_t cout = c0 >> 31;
      const uint64_t full = __ballot(g == 0xFFFFFFFFu);
      if (full) {  // a word of ones hands its carry-in on: resolve those in lane order
        uint64_t m = full;
        while (m) {
          const int l = __builtin_ctzll(m);
          const uint32_t cin_l = l ? (uint32_t)__builtin_amdgcn_readlane((int)cout, l - 1) : 0u;
          if (lane == l) cout = cin_l;  // (32 ones: the last carry equals the carry-in)
          m &= m - 1;
        }
      }
      const uint32_t cin = from_prev(cout);
      const uint32_t low = g & ~(g + 1u);              // the run of ones at bit 0
      const uint32_t c = c0 ^ (low & (0u - cin));       // an entering carry flips that run's pattern
      const uint32_t cl = (c << 1) | cin;               // carry INTO every column
      const uint32_t x = ~(a | cl) & colmask;
      if (TAG)
        __hip_atomic_store(tp + (size_t)(jb + r) * 64, ((uint64_t)generation << 32) | x, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      else
        ap[(size_t)(jb + r) * 64] = x;
      // marks for the row below
      const uint32_t ml = x & D[r][1], md = x & D[r][2], mr = x & D[r][3];
      a = ((ml >> 1) | (from_next(ml) << 31)) | md | ((mr << 1) | (from_prev(mr) >> 31));
      if (!TAG) sp[(size_t)(jb + r + 1) * 64] = a;
    }
  };
  uint32_t P[PF][4], Q[PF][4];
  load_batch(P, first);
  for (int j0 = 0; j0 < nrows; j0 += 2 * PF) {
    load_batch(Q, first + j0 + PF);
    run_batch(P, first + j0);
    if (j0 + PF >= nrows) break;
    load_batch(P, first + j0 + 2 * PF);
    run_batch(Q, first + j0 + PF);
  }
}

// ---- reach 2 and 3: the same by iteration inside the row (formulation: k_commit.h)
template <int R>
struct NlkCommitRows {
  static constexpr int side = 2 * R + 1, NP = R + R * side;
  static constexpr int PF = 48 / NP;  // rows per batch of planes in flight (two register sets)
};

// astate[(j * R + k) * 64 + word]: the marks rows j-1, j-2, .. have left for row j + k when row j starts.
// TAG: decisions as generation-tagged words for consumers that poll (see nlk_commit_rows1), whole grid only.
template <int R, bool TAG>
__device__ __forceinline__ void nlk_commit_rows(const uint32_t* __restrict__ planes, uint32_t* __restrict__ actbits,
                                                uint32_t* __restrict__ astate, uint64_t* __restrict__ tagged,
                                                uint32_t generation, int ngx, int first, int nrows, int lane) {
  constexpr int side = 2 * R + 1, NP = R + R * side, PF = NlkCommitRows<R>::PF;
  const int nb = ngx - 32 * lane;
  const uint32_t colmask = nb <= 0 ? 0u : (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u));
  auto from_prev = [](uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
  };
  auto from_next = [](uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
  };
  // whole-row shifts towards higher / lower columns by 1 <= d <= R bits
  auto up = [&](uint32_t v, int d) { return __builtin_amdgcn_alignbit(v, from_prev(v), 32 - d); };
  auto down = [&](uint32_t v, int d) { return __builtin_amdgcn_alignbit(from_next(v), v, d); };
  const uint32_t* pp = planes + lane;
  uint32_t* ap = TAG ? nullptr : actbits + lane;
  uint32_t* sp = TAG ? nullptr : astate + lane;
  uint64_t* tp = TAG ? tagged + lane : nullptr;
  uint32_t A[R];
#pragma unroll
  for (int k = 0; k < R; ++k) A[k] = (!TAG && first) ? sp[((size_t)first * R + k) * 64] : 0u;
  auto load_batch = [&](uint32_t (&D)[PF][NP], int jb) {
    const uint32_t* q = pp + (size_t)jb * NP * 64;
#pragma unroll
    for (int r = 0; r < PF; ++r)
#pragma unroll
      for (int p = 0; p < NP; ++p) D[r][p] = q[(r * NP + p) * 64];
  };
  auto run_batch = [&](const uint32_t (&D)[PF][NP], int jb) {
#pragma unroll
    for (int r = 0; r < PF; ++r) {
      if (jb + r >= first + nrows) break;  // (no row of the next band is replayed from planes not written yet)
      const uint32_t n = ~A[0] & colmask;
      uint32_t x = n;
      for (;;) {
        uint32_t blocked = 0;
#pragma unroll
        for (int d = 1; d <= R; ++d) blocked |= up(x & D[r][d - 1], d);
        const uint32_t y = n & ~blocked;
        const bool changed = y != x;
        x = y;
        if (!__ballot(changed)) break;
      }
      if (TAG)
        __hip_atomic_store(tp + (size_t)(jb + r) * 64, ((uint64_t)generation << 32) | x, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      else
        ap[(size_t)(jb + r) * 64] = x;
#pragma unroll
      for (int dj = 1; dj <= R; ++dj) {
        uint32_t cb = x & D[r][R + (dj - 1) * side + R];  // di = 0
#pragma unroll
        for (int di = 1; di <= R; ++di) {
          cb |= up(x & D[r][R + (dj - 1) * side + R + di], di);
          cb |= down(x & D[r][R + (dj - 1) * side + R - di], di);
        }
        A[dj - 1] = (dj < R ? A[dj] : 0u) | cb;
        if (!TAG) sp[((size_t)(jb + r + 1) * R + dj - 1) * 64] = A[dj - 1];
      }
    }
  };
  uint32_t P[PF][NP], Q[PF][NP];
  load_batch(P, first);
  for (int j0 = 0; j0 < nrows; j0 += 2 * PF) {
    load_batch(Q, first + j0 + PF);
    run_batch(P, first + j0);
    if (j0 + PF >= nrows) break;
    load_batch(P, first + j0 + 2 * PF);
    run_batch(Q, first + j0 + PF);
  }
}
