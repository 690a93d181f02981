// k_commit.h — replay of the reference's raster-order "processed" mask
// (reference: src/nlkalman.c:597-600 skip test, :930-931 mark; smoother
// :1490-1493, :1843-1844).
//
// In the reference a target is skipped when an EARLIER, non-skipped target's
// group contained it. Whether a target is skipped therefore depends only on
// integer records (group coordinates, np0) that the matching kernel already
// produced for every target, not on any filtered pixel. Five replays, all exact:
//   k_mask_commit_rows1   reach 1 (the steady state of the pipelines): one grid ROW per step on bit
//                         planes, a carry chain per row solved word-parallel (below)
//   k_mask_commit_rows<R> reach 2 and 3 (first frames of 8 x 8 patches, 12 x 12 patches): one grid row per step, the
//                         chain inside the row solved by iteration to its fixed point
//   k_mask_commit_wave<R> reach <= 3: anti-diagonal wavefront, lane = grid row, marks handed down by DPP (grids wider
//                         than 2048 targets, NLK_COMMIT_WAVE=1)
//   k_mask_commit<RPT>    the same wavefront with the mask as an LDS bitmap (first version, kept
//                         for comparison)
//   k_mask_commit_lists   any reach, from the group-coordinate lists
// The wavefront replays: a marking group reaches at most R grid cells, so target (i, j) can be decided
// at time i + (R+1)*j, when every target that can mark it has already been decided. k_mask_commit:
// one workgroup, one thread per grid row (RPT rows when the grid has more than 1024 rows), one barrier
// per time step.
//
// The mask lives in LDS as one bit per grid target (only grid-aligned
// coordinates are ever tested). Only FORWARD marks (targets later in raster
// order) are applied: a bit is then only ever set by targets decided before
// its owner, so the final bit array IS the skip decision and nothing is
// written to HBM inside the loop.
#pragma once
#include "nlk_common.h"
#include "k_commit_rows.h"

template <int RPT>  // grid rows per thread: row j = threadIdx.x + r * blockDim.x
__global__ void __launch_bounds__(1024)
k_mask_commit(const uint64_t* __restrict__ marks, uint8_t* __restrict__ active, int ngx,
              int ngy, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t* bits = (uint32_t*)smem;
  const int ntot = ngx * ngy;
  // + a tail: marks aimed below the last grid row (strip mode) land in padding
  const int nwords = (ntot + (R + 1) * ngx + 63) / 32 + 1;
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) bits[i] = 0;
  __syncthreads();
  const int side = 2 * R + 1;
  const int skew = R + 1;
  const int nsteps = ngx + skew * (ngy - 1);
  auto fetch = [&](int j, int s) -> uint64_t {
    const int i = s - skew * j;
    return (j < ngy && i >= 0 && i < ngx) ? marks[(size_t)j * ngx + i] : 0ull;
  };
  // Decide the targets of time step s; mw = their mark words. Bit
  // (dj+R)*side + (di+R) of a mark word stands for the grid neighbour (di, dj);
  // the raster-forward neighbours are exactly the bits above the centre bit. Each
  // forward row of (up to) `side` neighbours is OR-ed into the bit array with at
  // most two 32-bit LDS atomics: no per-bit loop, no division.
  const int centre = R * side + R;
  const uint32_t rowmask = (1u << side) - 1u;
  // LDS atomics retire about one lane per clock on gfx950, so they are issued
  // only by the lanes that really have a bit to set.
  auto or_bits = [&](int pos, uint32_t mask) {  // bits[pos ...] |= mask
    const int wd = pos >> 5, sh = pos & 31;
    const uint64_t m2 = (uint64_t)mask << sh;
    if ((uint32_t)m2) atomicOr(&bits[wd], (uint32_t)m2);
    if ((uint32_t)(m2 >> 32)) atomicOr(&bits[wd + 1], (uint32_t)(m2 >> 32));
  };
  auto decide = [&](int s, const uint64_t (&mw)[RPT]) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int j = threadIdx.x + r * blockDim.x;
      const int i = s - skew * j;
      const bool in = j < ngy && i >= 0 && i < ngx;
      const int t = in ? j * ngx + i : 0;
      const bool done = (bits[t >> 5] >> (t & 31)) & 1u;
      const uint64_t fwd = (in && !done) ? (mw[r] >> (centre + 1)) : 0ull;
      or_bits(t + 1, (uint32_t)fwd & ((1u << R) - 1u));          // same row, di = 1..R
      for (int dj = 1; dj <= R; ++dj)                              // rows below, di = -R..R
        or_bits(t + dj * ngx - R, (uint32_t)(fwd >> (R + (dj - 1) * side)) & rowmask);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS atomics issued; loads stay in flight
    __builtin_amdgcn_s_barrier();
  };
  // The mark words of a row are consumed one per step. They are streamed in
  // phases of S steps: while phase p is decided from registers A, the loads of
  // phase p+1 into B are in flight, so no HBM latency sits between barriers.
  constexpr int S = 16;
  uint64_t A[RPT][S], B[RPT][S];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int e = 0; e < S; ++e) A[r][e] = fetch(threadIdx.x + r * blockDim.x, e);
  for (int s0 = 0; s0 < nsteps; s0 += S) {
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int e = 0; e < S; ++e) B[r][e] = fetch(threadIdx.x + r * blockDim.x, s0 + S + e);
#pragma unroll
    for (int e = 0; e < S; ++e) {
      uint64_t mw[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) mw[r] = A[r][e];
      decide(s0 + e, mw);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int e = 0; e < S; ++e) A[r][e] = B[r][e];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ntot; t += blockDim.x)
    active[t] = !((bits[t >> 5] >> (t & 31)) & 1u);
}

// ---------------------------------------------------------------------------
// Register-resident variant (patch grids of up to 1024 rows): no LDS mask, no
// barrier. Lane = grid row; the marks a row receives from the R rows above are
// 2R+1-bit row masks that travel lane -> lane+1 by DPP (wave_shr:1) and are
// OR-ed into a small per-lane shift register of pending marks for the next
// (R+1)^2 columns; a row's own forward marks go into the same register. Only
// the last R lanes of a wavefront publish their masks through an LDS ring, and
// the next wavefront (which runs one phase of 16 steps behind) polls a progress
// word once per phase before consuming them. A step is 13 branch-free instructions on
// 32-bit registers (the forward bits of the mark words arrive pre-shifted and ordered by step from
// k_marks_skew) instead of an LDS round trip + atomics + workgroup barrier.
// ---------------------------------------------------------------------------
#define NLK_CW_RING 128  // steps of edge data kept per wavefront
#ifndef NLK_CW_PHASE
#define NLK_CW_PHASE 32
#endif

// Pre-pass: the forward part of every mark word (the bits after the target itself: same row,
// then the rows below; at most R + R*side <= 24 bits), stored by STEP of the replay:
// skewed[(i + (R+1)*j) * rows + j]. Lane j of k_mask_commit_wave then reads word s*rows + j at
// step s: one coalesced, unconditional 256-byte load per wavefront instead of 64 cache lines
// under a branch (entries outside the grid stay 0 and mark nothing).
// Grids of more than 1024 rows are replayed in bands of rows: a band starts with the last R rows of
// the previous one as context (`ctx_rows`), whose decisions are known (`ctx_active`): the marks of
// their skipped targets are dropped here, so that whatever the replay decides for the context rows,
// exactly the marks of the truly active ones reach the rows below (an active target can only be
// covered by an active earlier target, and those inside the context are all judged active).
template <int R>
__global__ void k_marks_skew(const uint64_t* __restrict__ marks, uint32_t* __restrict__ skewed, int ngx,
                             int ngy, int rows, const uint8_t* __restrict__ ctx_active, int ctx_rows) {
  constexpr int side = 2 * R + 1, centre = R * side + R;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ngx * ngy) return;
  const int j = t / ngx, i = t - j * ngx;
  uint32_t fwd = (uint32_t)(marks[t] >> (centre + 1));
  if (j < ctx_rows && !ctx_active[t]) fwd = 0;
  skewed[(size_t)(i + (R + 1) * j) * rows + j] = fwd;
}

template <int R>
__global__ void __launch_bounds__(1024)
k_mask_commit_wave(const uint32_t* __restrict__ skewed, uint8_t* __restrict__ active, int ngx,
                   int ngy, int skip_rows /* context rows of a band: decided before, not stored */) {
  constexpr int side = 2 * R + 1, skew = R + 1, centre = R * side + R;
  constexpr uint32_t rowmask = (1u << side) - 1u;
  constexpr int S = NLK_CW_PHASE;  // steps per phase: progress is exchanged once per phase
  __shared__ uint32_t edge[16][NLK_CW_RING][R];  // [wave][step % ring][lane 64-R+e]: packed row masks
  __shared__ int prog[16];                        // steps published by each wavefront
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int j = threadIdx.x;  // grid row
  if (lane == 0) prog[wave] = 0;
  __syncthreads();
  const int nsteps = ngx + skew * (ngy - 1);
  // (blockDim.x = rows of the skewed array; it holds 2 phases of zero rows past the last step)
  auto fetch = [&](int s) -> uint32_t { return skewed[(size_t)s * blockDim.x + j]; };
  auto ld_prog = [&](int w) {
    // acquire: the edge[] words read after a successful poll were written before the matching release
    return __hip_atomic_load(&prog[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  uint32_t A[S], B[S];
#pragma unroll
  for (int e = 0; e < S; ++e) A[e] = fetch(e);
  uint32_t pend = 0;  // bit b: column (current + b) of this row is already marked
  uint32_t outp = 0;  // row masks this lane produced in the previous step, `side` bits per dj
  for (int s0 = 0; s0 < nsteps; s0 += S) {
#pragma unroll
    for (int e = 0; e < S; ++e) B[e] = fetch(s0 + S + e);
    // ---- masks the previous wavefront's last R lanes produced in steps s0-1 .. s0+S-2:
    // wait once per phase until it has published them, then read them in one go
    uint32_t above[S];
#pragma unroll
    for (int e = 0; e < S; ++e) above[e] = 0;
    if (wave > 0) {
      while (ld_prog(wave - 1) < min(s0 + S - 1, nsteps)) __builtin_amdgcn_s_sleep(1);
      if (lane < R) {
#pragma unroll
        for (int e = 0; e < S; ++e) {
          const int sp = s0 + e - 1;  // producing step
          uint32_t acc = 0;
#pragma unroll
          for (int dj = 1; dj <= R; ++dj)  // dj > lane: source lane 64 + lane - dj of the previous wavefront
            if (dj > lane && sp >= 0)
              acc |= ((edge[wave - 1][sp % NLK_CW_RING][R + lane - dj] >> ((dj - 1) * side)) & rowmask)
                     << (skew * dj - 1 - R);
          above[e] = acc;
        }
      }
    }
    // flow control: the slots written in this phase must have been consumed
    if (wave + 1 < nwaves)
      while (ld_prog(wave + 1) < s0 + S - NLK_CW_RING + 1) __builtin_amdgcn_s_sleep(1);

    uint32_t flags[S / 4];  // decisions of this phase, one byte each
#pragma unroll
    for (int q = 0; q < S / 4; ++q) flags[q] = 0;
#pragma unroll
    for (int e = 0; e < S; ++e) {
      const int s = s0 + e;
      // ---- marks produced one step ago by the R rows above (inside this wavefront by DPP;
      // wave_shr zero-fills, so the first lanes receive nothing from beyond the wavefront)
      uint32_t sh = outp, inc = above[e];
#pragma unroll
      for (int dj = 1; dj <= R; ++dj) {
        sh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sh, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
        inc |= ((sh >> ((dj - 1) * side)) & rowmask) << (skew * dj - 1 - R);
      }
      pend |= inc;
      // ---- decide this row's target of step s: pure bit arithmetic, no compare / select
      // (outside the grid the mark word is 0 and the decision is never stored)
      const uint32_t act = ~pend & 1u;
      flags[e / 4] |= act << (8 * (e % 4));
      const uint32_t fwd = A[e] & (0u - act);
      outp = (fwd >> R) & ((1u << (R * side)) - 1u);  // rows below, side bits per dj
      pend = (pend >> 1) | (fwd & ((1u << R) - 1u));  // same row, columns i+1 .. i+R
      if (lane >= 64 - R) edge[wave][s % NLK_CW_RING][lane - (64 - R)] = outp;
    }
    // release: this phase's edge words (plain LDS stores above) are visible before the progress word
    if (lane == 63)
      __hip_atomic_store(&prog[wave], s0 + S, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    // write the phase's decisions: whole 32-bit words where the 4 columns exist
    {
      const int i0 = s0 - skew * j;  // column of step s0 for this row
      uint8_t* row = active + (size_t)j * ngx;
#pragma unroll
      for (int q = 0; q < S / 4; ++q) {
        const int i = i0 + 4 * q;
        if (j < skip_rows) {
        } else if (j < ngy && i >= 0 && i + 3 < ngx) {
          __builtin_memcpy(row + i, &flags[q], 4);
        } else if (j < ngy) {
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (i + b >= 0 && i + b < ngx) row[i + b] = (uint8_t)(flags[q] >> (8 * b));
        }
      }
    }
#pragma unroll
    for (int e = 0; e < S; ++e) A[e] = B[e];
  }
}


// ---------------------------------------------------------------------------
// Reach 1 (8 x 8 patches with the default temporal radius: the steady state of the pipelines): the
// replay one grid ROW per step instead of one diagonal per step.
//
// With R = 1 a target (i, j) is skipped iff it was marked by an active target of the row above
// (columns i-1, i, i+1) or by its active left neighbour. Row j-1 is final when row j starts, so the marks
// from above are three bit-plane ANDs and two one-bit shifts of whole rows: a = marked from above. Inside
// the row, with x = active, ms = "marks its right neighbour": x_i = !a_i & !(x_{i-1} & ms_{i-1}); the
// carry c_i = x_i & ms_i obeys c_i = g_i & !c_{i-1} with g = ms & ~a, i.e. inside every run of ones of g
// the carries alternate 1, 0, 1, ... from the run's first bit. Bits at an even distance from the start of
// their run are picked word-parallel with the run-start / add-carry trick (adding the even-positioned run
// starts to g clears exactly the runs that start on even bits). A row of up to 2048 targets is 64 lanes
// x 32 bits: 1-bit shifts across lanes by DPP wave_shr / wave_shl; a run that crosses a word boundary
// enters the next word as its carry-in (one DPP move: a word's carry-out depends on its carry-in only if
// the word is ALL ones, which takes the serial fix-up loop below). ~35 dependent vector instructions per
// row of the grid instead of ~20 per target of its longest diagonal: 269 steps instead of 1015 at 1080p.
// ---------------------------------------------------------------------------

// bit planes of the mark words of every grid row: planes[(j * 4 + p) * 64 + word], p = 0: marks (i+1, j),
// 1..3: marks (i-1, j+1), (i, j+1), (i+1, j+1); bit b of word w = column 32 w + b. All 64 words of a row
// are written (zeros past the grid), launched over 2048 columns.
__global__ void __launch_bounds__(256)
k_marks_planes1(const uint64_t* __restrict__ marks, uint32_t* __restrict__ planes, int ngx, int j0,
                uint32_t* __restrict__ generation = nullptr) {  // (the in-launch replay's counter: advanced here, never 0)
  constexpr int R = 1, side = 3, centre = R * side + R;
  if (generation && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const uint32_t v = *generation + 1u;
    *generation = v ? v : 1u;
  }
  const int j = j0 + blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;  // (rows j0 .. : a band of the grid)
  const uint32_t fwd = i < ngx ? (uint32_t)(marks[(size_t)j * ngx + i] >> (centre + 1)) : 0u;
  const int lane = threadIdx.x & 63, w0 = i >> 5;  // (i of lane 0 of the wavefront is a multiple of 64)
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint64_t b = __ballot((fwd >> p) & 1u);
    if (lane == 0) *(uint64_t*)(planes + ((size_t)j * 4 + p) * 64 + w0) = b;
  }
}

// One wavefront; lane = word of the row. A lone wavefront issues an instruction every ~6 cycles whatever
// it is, so the loop is written for instruction count: fixed row strides (immediate offsets, no
// predicates: planes and decisions are padded to whole batches), two register sets for the planes in
// flight that swap roles (no copies).
__global__ void __launch_bounds__(64)
k_mask_commit_rows1(const uint32_t* __restrict__ planes, uint32_t* __restrict__ actbits,
                    uint32_t* __restrict__ astate, int ngx, int first, int nrows) {
  nlk_commit_rows1<NLK_CR_BATCH, false>(planes, actbits, astate, nullptr, 0u, ngx, first, nrows, threadIdx.x);
}

// the decision bits of tagged words -> the byte per target (records, on request: nlk_ctx_read_records)
__global__ void __launch_bounds__(256)
k_active_bytes_tagged(const uint64_t* __restrict__ tagged, uint8_t* __restrict__ active, int ngx) {
  const int j = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < ngx) active[(size_t)j * ngx + i] = ((uint32_t)tagged[(size_t)j * 64 + (i >> 5)] >> (i & 31)) & 1u;
}

// ---------------------------------------------------------------------------
// Reach 2 and 3 (spatial first frames of 8 x 8 patches, 12 x 12 patches): the same replay by grid ROWS.
// A target (i, j) is skipped iff an active target of the R rows above marked it (columns i-R .. i+R: whole-row
// bit-plane ANDs and shifts, accumulated per row below in A[1..R]), or one of its R left neighbours did:
//   x_i = n_i & !(x_{i-1} & m1_{i-1}) & ... & !(x_{i-R} & mR_{i-R}),   n = not marked from above.
// The system is triangular (x_i depends on columns to its left only), so the Jacobi iteration
// x <- n & ~(shl(x & m1, 1) | ... | shl(x & mR, R)) from x = n reaches THE solution, and has reached it when
// an iteration changes nothing; column i is final after at most i+1 iterations, in practice after the length
// of the longest chain of left-marking neighbours (about ten at 1080p). 269 row steps of ~36 + 12 per
// iteration vector instructions instead of 1015 diagonal steps of ~20.
// planes[(j * NP + p) * 64 + word], NP = R + R (2R+1): p < R: marks (i+p+1, j); then dj = 1..R, di = -R..R.
// ---------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(256)
k_marks_planes(const uint64_t* __restrict__ marks, uint32_t* __restrict__ planes, int ngx, int j0,
               uint32_t* __restrict__ generation = nullptr) {
  constexpr int side = 2 * R + 1, centre = R * side + R, NP = R + R * side;
  if (generation && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const uint32_t v = *generation + 1u;
    *generation = v ? v : 1u;
  }
  const int j = j0 + blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t fwd = i < ngx ? (uint32_t)(marks[(size_t)j * ngx + i] >> (centre + 1)) : 0u;
  const int lane = threadIdx.x & 63, w0 = i >> 5;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint64_t b = __ballot((fwd >> p) & 1u);
    if (lane == 0) *(uint64_t*)(planes + ((size_t)j * NP + p) * 64 + w0) = b;
  }
}

template <int R>
__global__ void __launch_bounds__(64)
k_mask_commit_rows(const uint32_t* __restrict__ planes, uint32_t* __restrict__ actbits,
                   uint32_t* __restrict__ astate, int ngx, int first, int nrows) {
  nlk_commit_rows<R, false>(planes, actbits, astate, nullptr, 0u, ngx, first, nrows, threadIdx.x);
}

// bits -> the byte per target the group kernels read
__global__ void __launch_bounds__(256)
k_active_bytes(const uint32_t* __restrict__ actbits, uint8_t* __restrict__ active, int ngx, int j0) {
  const int j = j0 + blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < ngx) active[(size_t)j * ngx + i] = (actbits[(size_t)j * 64 + (i >> 5)] >> (i & 31)) & 1u;
}

// ---------------------------------------------------------------------------
// Any reach (R > 3: (2R+1)^2 neighbours do not fit a 64-bit mark word): the same time-stepped
// replay driven by the group-coordinate lists themselves. One workgroup; thread = grid row (rows
// tid, tid + 1024, ...); at step s row j looks at column s - (R+1) j. `active` is the mask: 1 until
// an earlier, non-skipped target's marking group contains the target (only raster-FORWARD members
// are written, so the final array is the decision). One barrier per step and up to nagg dependent
// stores per target: slow (milliseconds per frame), but it is the path of unusual parameter
// combinations only (e.g. --f1_p 4 with the default search radius 10, or --f1_st 20).
// Bytes are exchanged between the threads through agent-scope atomics (no stale L1 lines).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_mask_commit_lists(const NlkTarget* __restrict__ tinfo, const uint32_t* __restrict__ gcoords, int gstride,
                    uint8_t* __restrict__ active, int ngx, int ngy, int R, int step, int oy) {
  const int skew = R + 1;
  const int nsteps = ngx + skew * (ngy - 1);
  const int ntot = ngx * ngy;
  for (int t = threadIdx.x; t < ntot; t += blockDim.x)
    __hip_atomic_store(&active[t], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    for (int j = threadIdx.x; j < ngy; j += blockDim.x) {
      const int i = s - skew * j;
      if (i < 0 || i >= ngx) continue;
      const int t = j * ngx + i;
      if (!__hip_atomic_load(&active[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
      const NlkTarget info = tinfo[t];
      if (!(info.flags & 2)) continue;  // this group does not mark (reference: :931, smoother :1844)
      const int px = i * step, py = oy + j * step;
      for (int n = 0; n < info.nagg; ++n) {
        const uint32_t q = gcoords[(size_t)t * gstride + n];
        const int dx = nlk_x(q) - px, dy = nlk_y(q) - py;
        if (dx % step || dy % step) continue;
        const int ii = i + dx / step, jj = j + dy / step;
        if (ii < 0 || ii >= ngx || jj < 0 || jj >= ngy) continue;
        const int t2 = jj * ngx + ii;
        if (t2 > t) __hip_atomic_store(&active[t2], (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
}
