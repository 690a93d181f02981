// k_match.h — exhaustive block matching + ranked k-NN selection + group
// membership, for every target of the patch grid
// (reference: src/nlkalman.c:605-609, 630-707, 725-732, 779-793, 857, 931).
//
// One workgroup owns a tile of TGX x TGY grid targets and stages the part of
// the matching image those targets can reach (patches + search halo) in LDS,
// planar per channel, with coalesced row reads from HBM. One wavefront
// processes one target at a time: lane = candidate. The squared distance is
// accumulated in the reference's element order (hy, hx, c) with one rounding
// per multiply and per add, so that the ranking is reproducible bit for bit by
// the CPU oracle. Selection is by rank counting on the (distance, window index)
// key, which is exactly a stable ascending sort's prefix.
#pragma once
#include "nlk_common.h"

#define NLK_BM_THREADS 256
#define NLK_BM_WAVES (NLK_BM_THREADS / 64)

struct NlkTile {
  int tgx, tgy;      // targets per tile
  int ntx, nty;      // tiles
  int rw_max, rh_max; // LDS region capacity (floats per row / rows)
  int ncand_max;     // capacity of the per-wave distance / selection arrays
};

__device__ inline uint64_t nlk_wave_or(uint64_t v) {
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
    v |= ((uint64_t)hi << 32) | lo;
  }
  return v;
}

template <int MAXM>
__global__ void __launch_bounds__(NLK_BM_THREADS)
k_bm_topk(const float* __restrict__ img, const uint8_t* __restrict__ vmap, NlkGeom g,
          NlkTile tl, uint32_t* __restrict__ topk, NlkTarget* __restrict__ tinfo,
          uint32_t* __restrict__ gcoords, uint64_t* __restrict__ marks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile_x = blockIdx.x % tl.ntx, tile_y = blockIdx.x / tl.ntx;
  const int gx0 = tile_x * tl.tgx, gy0 = tile_y * tl.tgy;
  const int cx = min(tl.tgx, g.ngx - gx0), cy = min(tl.tgy, g.ngy - gy0);
  const int wmax = g.smoother ? g.wsz_t
                              : (g.have_prev ? max(g.wsz_x, g.wsz_t) : g.wsz_x);

  // image region reachable from this tile
  const int rx0 = max(gx0 * g.step - wmax, 0);
  const int rx1 = min((gx0 + cx - 1) * g.step + wmax + g.psz, g.w);
  const int ry0 = max(g.oy + gy0 * g.step - wmax, 0);
  const int ry1 = min(g.oy + (gy0 + cy - 1) * g.step + wmax + g.psz, g.h);
  const int rw = rx1 - rx0, rh = ry1 - ry0;
  const int rwp = tl.rw_max;  // padded row stride (odd)
  const int plane = rwp * tl.rh_max;

  float* tile = smem;                                   // [ch][rh_max][rwp]
  uint32_t* dist_all = (uint32_t*)(tile + g.ch * plane); // [waves][ncand_max]
  uint32_t* sel_all = dist_all + NLK_BM_WAVES * tl.ncand_max;
  uint32_t* grp_all = sel_all + NLK_BM_WAVES * tl.ncand_max; // [waves][gstride]

  const size_t npix = (size_t)g.w * g.h;
  for (int c = 0; c < g.ch; ++c)
    for (int i = threadIdx.x; i < rw * rh; i += NLK_BM_THREADS) {
      const int y = i / rw, x = i - y * rw;
      tile[c * plane + y * rwp + x] = img[c * npix + (size_t)(ry0 + y) * g.w + rx0 + x];
    }
  __syncthreads();

  uint32_t* dl = dist_all + wave * tl.ncand_max;
  uint32_t* sel = sel_all + wave * tl.ncand_max;
  uint32_t* grp = grp_all + wave * g.gstride;
  const float norm = (float)g.psz * g.psz * g.ch;

  for (int tt = wave; tt < cx * cy; tt += NLK_BM_WAVES) {
    const int ty = tt / cx, tx = tt - ty * cx;
    const int gx = gx0 + tx, gy = gy0 + ty;
    const int px = gx * g.step, py = g.oy + gy * g.step;
    const size_t t = (size_t)gy * g.ngx + gx;
    const int prev_p = g.have_prev ? vmap[(size_t)py * g.w + px] : 0;
    int k = prev_p ? g.npt : g.npx;
    NlkTarget info = {0, 0, 0, prev_p};
    if (k <= 1) {
      // single-patch mode aggregates nothing in the filter (reference: :815-857);
      // the smoother passes the target patch through (reference: :1795-1804)
      if (g.smoother) {
        info.nagg = 1;
        if (lane == 0) gcoords[t * g.gstride] = nlk_pack_xy(px, py);
      }
      if (lane == 0) { tinfo[t] = info; marks[t] = 0; }
      continue;
    }
    const int wsz = (g.smoother || prev_p) ? g.wsz_t : g.wsz_x;
    const int x0 = max(px - wsz, 0), x1 = min(px + wsz, g.w - g.psz) + 1;
    const int y0 = max(py - wsz, 0), y1 = min(py + wsz, g.h - g.psz) + 1;
    const int nwx = x1 - x0, n = nwx * (y1 - y0);
    k = min(k, n);

    // --- distances: lane owns candidates lane, lane+64, ...
    int cq[MAXM];    // LDS offset of the candidate's origin inside a plane
    float acc[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      const int i = min(lane + 64 * m, n - 1);
      const int wy = i / nwx, wx = i - wy * nwx;
      cq[m] = (y0 + wy - ry0) * rwp + (x0 + wx - rx0);
      acc[m] = 0.f;
    }
    const int tq = (py - ry0) * rwp + (px - rx0);
    // one rounding per subtract, multiply and add (no FMA contraction): the
    // CPU restatement is built with -ffp-contract=off and must rank identically
    for (int hy = 0; hy < g.psz; ++hy)
      for (int hx = 0; hx < g.psz; ++hx) {
        const int o = hy * rwp + hx;
        for (int c = 0; c < g.ch; ++c) {
#pragma clang fp contract(off)
          const float tv = tile[c * plane + tq + o];
#pragma unroll
          for (int m = 0; m < MAXM; ++m) {
            if (m * 64 < n) {  // wave-uniform
              const float e = tile[c * plane + cq[m] + o] - tv;
              const float e2 = e * e;
              acc[m] = acc[m] + e2;
            }
          }
        }
      }
    uint32_t dk[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      const float q = __fdiv_rn(acc[m], norm);
      dk[m] = __float_as_uint(q > 0.f ? q : 0.f);
      if (lane + 64 * m < n) dl[lane + 64 * m] = dk[m];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS writes of this wave landed
    __builtin_amdgcn_wave_barrier();

    // --- rank of every candidate under the (distance, index) order
    int rank[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) rank[m] = 0;
    for (int j = 0; j < n; ++j) {
      const uint32_t dj = dl[j];
#pragma unroll
      for (int m = 0; m < MAXM; ++m)
        if (m * 64 < n)
          rank[m] += (dj < dk[m]) || (dj == dk[m] && j < lane + 64 * m);
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      const int i = lane + 64 * m;
      if (i < n && rank[m] < k) {
        const int wy = i / nwx, wx = i - wy * nwx;
        sel[rank[m]] = nlk_pack_xy(x0 + wx, y0 + wy);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();

    // --- group membership: the first ntagg kept candidates that have a valid
    // previous patch, or (none valid) the first ntagg kept candidates
    int np0 = 0;
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
      const int slot = np0 + __popcll(b & ((1ull << lane) - 1ull));
      if (v && slot < g.ntagg) grp[slot] = q;
      np0 += __popcll(b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();

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
        if (mark && dx % g.step == 0 && dy % g.step == 0) {
          const int di = dx / g.step, dj = dy / g.step;
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
      tinfo[t] = info;
      marks[t] = mbits;
    }
    __builtin_amdgcn_wave_barrier();
  }
}
