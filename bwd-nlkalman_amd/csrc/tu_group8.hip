// tu_group8.hip — launcher of the 8x8 group kernels: k_group8m (matrix cores, default) and
// k_group8 (registers + DPP, NLK_GROUP_DPP=1)
#include "k_gather.h"
#include "k_group8.h"
#include "k_group8m.h"
#include "nlk_internal.h"

const void* nlk_group8m_ilp_kernel(int ch, bool smoother, int sep);  // tu_group8_ilp.hip

namespace {

template <int CH, bool SMO>
int launch_group8_t(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                    const float* prev, float* acc, const uint8_t* active) {
  constexpr int PSZ = 8;
  // psz 8 runs its DCTs on the matrix cores (k_group8m.h); NLK_GROUP_DPP selects the
  // register/DPP kernel (k_group8.h) for comparison
  const bool mfma = !nlk_set(c->sw.group_dpp);
  // the matrix-core kernel's DCTs: the Kronecker form on 16 x 16 products (rounds 2-4) or the separable form on
  // 4 x 4 blocks (round 5), by pass - NLK_GROUP_SEP = 0: Kronecker in both passes, 2: Kronecker statistics,
  // separable filter / inverse / aggregate (pass B), 6: separable in both; the default by mode is worked out below.
  // Round 5's measurements, when the separable pass A still loaded a cache line per lane:
  // Default by what was measured at 1080p RGB (profiles/r05_mode_times_1080p.txt, group kernel, Kronecker -> hybrid):
  // FLT1 temporal 0.796 -> 0.740 ms, FLT1 spatial 1.043 -> 0.978; FLT2 (one member per group) 0.612 -> 0.628 and the
  // smoother 1.281 -> 1.322 stay with the Kronecker form.
  // Single-channel frames (`r05_mode_times_1080p_gray_by_sep.txt`; the separable pass B packs 16 members per step
  // there): FLT1 temporal 0.514 (Kronecker) / 0.404 (2) / 0.393 (6), first frame 0.594 / 0.487 / 0.482, the smoother
  // 0.926 / 0.547 / 0.576, FLT2 0.315 / 0.331 / 0.358.
  // Round 6: the smoother's pass B reads the difference image previous - image (k_layout) - one row set per member
  // instead of two - and with half the cache lines to fetch the separable pass B wins there too: SMO1 group at 1080p
  // RGB 1.287 ms (Kronecker, two row sets) -> 1.234 (Kronecker, difference image) -> 1.101 (separable pass B).
  // Round 6, later: the separable pass A loads a row as two halves in two ADJACENT lanes (one cache line per pair
  // in each load: k_group8m.h, row_read) and the vector L1, which had kept it behind the Kronecker form, stops being
  // what it waits for. Group ms at 1080p, forms 0 / 2 / 6 (`r06_mode_times_1080p_by_sep.txt`, `..._gray_by_sep.txt`):
  // RGB FLT1 temporal 0.804 / 0.698 / 0.662, FLT2 0.607 / 0.612 / 0.576, first frame 1.125 / 0.894 / 0.894, smoother
  // 1.207 / 0.971 / 0.992; one channel FLT1 temporal 0.534 / 0.413 / 0.384, FLT2 0.316 / 0.344 / 0.304, first frame
  // 0.532 / 0.403 / 0.394, smoother 0.850 / 0.563 / 0.526. So: one channel 6 everywhere; RGB 6 for the filter's temporal
  // frames, the hybrid form for first frames and the smoother (Kronecker for those with fewer than 4 members).
  const int sep_default = g.ch == 1 ? 6 : (!g.smoother && prev) ? 6 : (g.ntagg < 4 ? 0 : 2);
  int sep = mfma ? nlk_or(c->sw.group_sep, sep_default) : 0;
  if (sep != 0 && sep != 6) sep = 2;
  // (k_group8m addresses every patch as planes base + a 32-bit byte offset: the call's images must lie in the
  // context's slab and the slab be smaller than 4 GiB - ~119 Mpixel of RGB; beyond that the packed-lane kernel)
  if (mfma) {
    const float* lo = (const float*)c->planes.p;
    const float* hi = lo + c->planes.cap / sizeof(float);
    const bool inside = img >= lo && img < hi && cur >= lo && cur < hi && (!prev || (prev >= lo && prev < hi));
    if (!inside || c->planes.cap >= ((size_t)1 << 32)) {
      if (c->rv.chase_words) return fail(c, NLK_EINVAL, "internal: mask replay handed to a group kernel that cannot run it");
      return nlk_launch_groupp_a(c, g, img, cur, prev, acc, active);
    }
  }
  if (mfma && SMO && prev && !(c->p_diff && cur == c->p_cur && prev == c->p_prev)) {
    // (the smoother's pass B reads the difference image the layout kernel made of exactly these two images)
    if (c->rv.chase_words) return fail(c, NLK_EINVAL, "internal: mask replay handed to a group kernel that cannot run it");
    return nlk_launch_groupp_a(c, g, img, cur, prev, acc, active);
  }
  if (c->rv.chase_words && (!mfma || c->deterministic))
    return fail(c, NLK_EINVAL, "internal: mask replay handed to a group kernel that cannot run it");
  if (c->deterministic && !mfma)
    return fail(c, NLK_EUNSUP, "deterministic aggregation is not available in the NLK_GROUP_DPP variant");
  // Deterministic mode runs a temporal frame's far-reaching (spatial-branch) groups in a second
  // launch whose tiles have the spatial halo, so that no member ever leaves its tile (k_group8.h)
  const bool split = c->deterministic && g.have_prev && !g.smoother && g.wsz_x > g.wsz_t;
  // the tiles of a pass: halo, targets per tile, LDS strides
  auto shape = [&](int pass) {
    NlkGTile tl{};
    tl.pbase = (const float*)c->planes.p;
    // (the smoother's pass B transforms previous - image: laid out once per call by k_layout; `cur` is what gets filtered)
    tl.diff = (SMO && prev) ? c->p_diff : nullptr;
    tl.split = split;
    tl.far = pass;
    tl.chase = c->rv.chase_words ? c->rv.chase_reach : 0;
    tl.chase_test_skip0 = nlk_set(c->sw.chase_test_skip0);
    tl.chase_gen = c->rv.chase_gen;
    tl.chase_row0 = c->rv.chase_row0;
    tl.chase_rows = c->rv.chase_rows;
    tl.chase_planes = c->rv.chase_planes;
    tl.chase_words = c->rv.chase_words;
    // LDS tile halo = reach of the dominant kind of group; without the split the rare spatial-branch
    // groups of a temporal frame that reach further fall back to HBM atomics
    tl.wmax = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
    if (pass == 1) tl.wmax = g.wsz_x;
    // Targets per wavefront (NLK_GTX / NLK_GTY override for experiments), measured (profiles/README.md):
    // 3 x 1 on a 1080p grid (group 1.040 ms; 4 x 1 1.049, 2 x 1 1.068: a 3-target row needs 28 floats of
    // row stride like a 2-target one, 11 KB of LDS instead of 13), 2 x 1 on smaller grids (more, shorter
    // workgroups fill the chip better: 720p 0.499 vs 0.508 ms, 640 x 480 0.194 vs 0.221), one target while
    // the tiles would not even fill the ~2560 wavefronts the chip holds (256 x 256). With a wide halo the
    // tile would leave LDS room for ~1 wavefront per SIMD: 2 x 1 at most then.
    auto tiles_with = [&](int t) { return (size_t)((g.ngx + t - 1) / t) * g.ngy; };
    const int tgx_fill = tiles_with(3) >= 30000 ? 3 : (tiles_with(2) >= 2560 ? 2 : 1);
    // (round 4, first frames at 1080p - the spatial halo, 2 wavefronts per SIMD whatever the tile: 1 x 2 targets
    // 1.029 ms, 3 x 1 1.043, 3 x 2 1.042, 1 x 3 1.052, 2 x 2 1.090, 2 x 1 1.113 (the default until then), 1 x 4 1.113,
    // 1 x 1 1.143, 4 x 1 1.164, 1 x 6 1.185)
    const bool wide_full = mfma && tl.wmax > 6 && tgx_fill == 3 && pass == 0;
    // (round 6, after the tile's strides shrank its LDS footprint - first frames at 1080p RGB: 2 x 2 targets 0.896 ms,
    // 2 x 3 0.933, 3 x 2 0.935, 1 x 2 0.939 (the default until then), 2 x 1 0.938, 1 x 3 0.953, 1 x 4 1.008; 4K RGB 3.50
    // against 3.74; one channel 3 x 2 0.430, 2 x 3 0.433, 2 x 2 0.445, 1 x 2 0.47-0.50: tools/sweep_gt_spatial.sh)
    tl.tgx = nlk_or(c->sw.gtx, wide_full ? (CH == 1 ? 3 : 2) : min(tgx_fill, tl.wmax > 6 ? 2 : 4));
    // (round 3, matrix-core kernel with the leaner pass A, 1080p: 3 x 2 targets 0.957 ms, 2 x 2 0.970, 3 x 1 0.976,
    // 4 x 2 1.10, 2 x 3 1.05, 3 x 3 1.04 - two target rows share the tile's vertical halo: 40 % fewer flushed bytes;
    // round 4, straight-line aggregation: 3 x 2 0.831, 2 x 2 0.851, 3 x 1 0.869, 3 x 3 0.891, 4 x 2 0.993)
    tl.tgy = nlk_or(c->sw.gty, (mfma && tgx_fill == 3 && tl.wmax <= 6) || wide_full ? 2 : 1);
    tl.ntx = (g.ngx + tl.tgx - 1) / tl.tgx;
    tl.nty = (g.ngy + tl.tgy - 1) / tl.tgy;
    tl.nty_full = tl.nty;
    tl.single = 0;
    if (mfma && (tl.tgy > 1 || tl.tgx > 1) && !c->deterministic) {
      // The launch ends when its last tile does, and a 3 x 2 tile takes 1/7 of a 1080p launch (21600 tiles on 3072
      // wavefront slots): the chip would drain for most of a tile's time. So the last ~0.8 slots' worth of tiles
      // hold one grid row (3 x 1), and the last ~0.9 slots' worth of targets go one per workgroup
      // (1080p: 16 + 6 of 269 grid rows; group 0.950 -> 0.900 ms. NLK_G8_TAIL / NLK_G8_SINGLE override)
      int tail = tl.tgy == 1 ? 0 : nlk_or(c->sw.g8_tail, (2560 + tl.ntx / 2) / tl.ntx);
      int single = nlk_or(c->sw.g8_single, (2880 + g.ngx / 2) / g.ngx);
      single = max(0, min(single, g.ngy / 8));
      tail = max(0, min(tail, g.ngy / 4));
      tl.single = single;
      const int rows = g.ngy - tl.single;
      tl.nty_full = (rows - tail) / tl.tgy;
      tl.nty = tl.nty_full + (rows - tl.nty_full * tl.tgy);
    }
    tl.nmain = mfma ? nlk_g8m_grid(tl.ntx, tl.nty) : 0;
    const int rw_max = (tl.tgx - 1) * g.step + 2 * tl.wmax + g.psz;
    tl.rh_max = (tl.tgy - 1) * g.step + 2 * tl.wmax + g.psz;
    if (mfma && sep != 0 && CH > 1) {
      // Separable pass B (round 6). A tile update is a ds_read_b32 / ds_write_b32 of lane
      // (column g4 = lane >> 4, plane spl = (lane >> 2) & 3, row si = lane & 3): the LDS serves a 4-byte access in
      // two groups of 32 lanes, {0-31} and {32-63}, on 32 banks (MI355X_MICROARCH.md, LDS) - and a half wavefront
      // holds all FOUR planes here, two columns and four rows. With the Kronecker layout's strides (below: plane
      // stride 16 mod 32) planes 0 / 2 and 1 / 3 of a half met on the same banks: a 2-way conflict on every one of
      // the 160 tile instructions of a target (SQ_LDS_BANK_CONFLICT 1.6e6 -> 3.1e7 per C2 launch when the
      // separable pass B came in round 5). Conflict-free: row stride = 2 (mod 4) - rows si give four different
      // multiples of 2, the two columns fill the odd banks - and plane stride = 8 (mod 16) (brute force over all
      // strides: tools/lds_banks.py). At 1080p the row stride is the region's own 26 floats (28 before) and the
      // workgroup's LDS falls from ten to nine 1280-byte pieces.
      tl.rwp = rw_max + ((2 - rw_max) % 4 + 4) % 4;
      tl.plane = tl.rwp * tl.rh_max;
      tl.plane += ((8 - tl.plane) % 16 + 16) % 16;
    } else if (mfma) {
      // Kronecker pass B (plane = lane >> 4, pixel (pi, pj) = lane & 15; also the 16-member steps of one-channel
      // frames, whose rounds touch two planes): one aggregation access = 4x4 pixels of each plane; row stride = 4
      // (mod 8) and plane stride = 16 (mod 32 banks) make the 32 lanes of a half wavefront hit 32 banks
      tl.rwp = rw_max + ((4 - rw_max) % 8 + 8) % 8;
      tl.plane = tl.rwp * tl.rh_max;
      tl.plane += ((16 - tl.plane) % 32 + 32) % 32;
    } else {
      tl.rwp = rw_max | 1;
      tl.plane = tl.rwp * tl.rh_max;
    }
    return tl;
  };
  const int npass = split ? 2 : 1;
  NlkGTile tls[2] = {shape(0), shape(split ? 1 : 0)};
  // Deterministic mode: slabs, "written" flags and tile-row counters of BOTH passes, sized from each pass's OWN
  // tiles before the first launch (growing a buffer frees it). (Round 3 sized the far pass with the near pass's
  // tile rows - half as many since the near tiles hold two grid rows - and its counters likewise: writes past
  // the end that the allocation slack hid; found with NLK_DEBUG_GUARD=1.)
  size_t slab_at[2] = {0, 0}, flag_at[2] = {0, 0}, cnt_at[2] = {0, 0};
  if (c->deterministic) {
    size_t need = 0, nflag = 0, ncnt = 0;
    for (int pass = 0; pass < npass; ++pass) {
      const size_t ntiles = (size_t)tls[pass].ntx * tls[pass].nty;
      slab_at[pass] = need; flag_at[pass] = nflag; cnt_at[pass] = ncnt;
      need += ntiles * (CH + 1) * tls[pass].plane;
      nflag += ntiles;
      ncnt += 1 + (size_t)tls[pass].nty;
    }
    const size_t cnt_off = (nflag + 15) & ~(size_t)15, cnt_bytes = sizeof(int) * ncnt;
    int rc;
    if ((rc = reserve(c, c->slab, sizeof(float) * need)) || (rc = reserve(c, c->tflag, cnt_off + cnt_bytes))) return rc;
    HIPCHK(c, hipMemsetAsync((uint8_t*)c->tflag.p + cnt_off, 0, cnt_bytes, c->rv.stream));
    for (int pass = 0; pass < npass; ++pass) {
      tls[pass].slab = (float*)c->slab.p + slab_at[pass];
      tls[pass].tflag = (uint8_t*)c->tflag.p + flag_at[pass];
      tls[pass].tcount = (int*)((uint8_t*)c->tflag.p + cnt_off) + cnt_at[pass];
    }
  }
  for (int pass = 0; pass < npass; ++pass) {
    const NlkGTile& tl = tls[pass];
#ifndef NLK_G8_LDS_PAD
#define NLK_G8_LDS_PAD 0  // (experiments: bytes of unused LDS per workgroup, to cut the occupancy)
#endif
    const size_t stash = !mfma ? 0 : sep == 0 ? nlk_g8_stash_floats<CH, 0>() : sep == 2 ? nlk_g8_stash_floats<CH, 2>() : nlk_g8_stash_floats<CH, 6>();
    const size_t lds = sizeof(float) * ((size_t)(CH + 1) * tl.plane + stash) + NLK_G8_LDS_PAD;
    if (lds > 160 * 1024) return fail(c, NLK_EUNSUP, "aggregation tile needs %zu bytes of LDS", lds);
    void (*kern)(const float*, const float*, const float*, const uint8_t*, NlkGeom, NlkGTile,
                 const uint32_t*, const NlkTarget*, const uint32_t*, const uint8_t*, const float*,
                 const float*, float*);
    kern = !mfma ? k_group8<CH, SMO> : sep == 0 ? k_group8m<CH, SMO, 0> : sep == 2 ? k_group8m<CH, SMO, 2> : k_group8m<CH, SMO, 6>;
    // (the same kernel compiled under the max-ilp scheduler, where that is the faster one: tu_group8_ilp.hip;
    // NLK_GROUP_ILP=0 keeps this unit's)
    if (mfma && nlk_or(c->sw.group_ilp, 1) != 0)
      if (const void* k2 = nlk_group8m_ilp_kernel(CH, SMO, sep)) kern = reinterpret_cast<decltype(kern)>(const_cast<void*>(k2));
    HIPCHK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    const float* basis = (const float*)c->tabs.p;
    hipLaunchKernelGGL(kern, dim3(mfma ? tl.nmain + ((tl.single * g.ngx + 7) / 8) * 8 : nlk_xcd_grid(tl.ntx * tl.nty)), dim3(64), lds, c->rv.stream, img, cur, prev,
                       (const uint8_t*)c->vmap.p, g, tl, (const uint32_t*)c->rv.topk,
                       (const NlkTarget*)c->rv.tinfo, (const uint32_t*)c->rv.gcoords,
                       active, basis, basis + PSZ * PSZ, acc);
    if (c->deterministic)
      nlk_launch_gather(c->rv.stream, acc, tl.slab, tl.tflag, tl.tcount, g, tl, CH + 1);
    HIPCHK(c, hipGetLastError());
  }
  return NLK_OK;
}

}  // namespace

int nlk_launch_group8(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                      float* acc, const uint8_t* active) {
#define NLK_FAST(C)                                                                  \
  if (g.ch == C)                                                                     \
    return g.smoother ? launch_group8_t<C, true>(c, g, img, cur, prev, acc, active)  \
                      : launch_group8_t<C, false>(c, g, img, cur, prev, acc, active);
  NLK_FAST(1) NLK_FAST(3)
#undef NLK_FAST
  return fail(c, NLK_EUNSUP, "%d channels not supported (1 or 3)", g.ch);
}
