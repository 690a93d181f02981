// nlk_internal.h — what the translation units of libnlk_hip.so share: the context, error helpers,
// scratch-buffer growth and the launchers each unit exports to nlk_hip.hip. (The library is built
// from several .hip files so that `make -j` compiles the big kernels in parallel.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nlk_hip.h"
#include "nlk_common.h"

struct NlkTvMail;   // k_tvl1.h
struct NlkTile;     // k_match.h
struct NlkGTile;    // k_group8.h

struct NlkBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* base = nullptr;  // what hipMalloc returned (NLK_DEBUG_GUARD: p sits at the END of the allocation)
};

// what a launcher works on: the stream and the per-target records of the patch-grid rows it is
// given (the whole grid, a strip, or one band of a frame processed in bands on two streams)
struct NlkRecView {
  hipStream_t stream = nullptr;
  uint32_t* topk = nullptr;      // [targets][kmax]
  NlkTarget* tinfo = nullptr;    // [targets]
  uint32_t* gcoords = nullptr;   // [targets][gstride]
  uint64_t* marks = nullptr;     // [targets]
  uint32_t* wide = nullptr;      // [0] = queue length, then the queued targets (k_bm_wide)
  // set by the frame call when the group kernel is to replay the processed mask itself (k_group8m, NlkGTile::chase)
  const uint32_t* chase_planes = nullptr;
  uint64_t* chase_words = nullptr;
  const uint32_t* chase_gen = nullptr;
  int chase_reach = 0, chase_row0 = 0, chase_rows = 0;
};

// The NLK_* environment switches (variants for comparison tests and experiments, none
// needed in production) are read ONCE, when a context is created - not on every frame call - and again only
// on request (nlk_ctx_reload_switches: the tests change the environment under a live context).
// A field holds atoi(value), or NLK_UNSET when the variable does not exist.
#define NLK_UNSET (-2147483647 - 1)
#define NLK_SWITCH_LIST(X)                                                                                   \
  X(deterministic, "NLK_DETERMINISTIC") X(generic_group, "NLK_GENERIC_GROUP") X(group_packed, "NLK_GROUP_PACKED") \
  X(group_dpp, "NLK_GROUP_DPP") X(group_sep, "NLK_GROUP_SEP") X(group_ilp, "NLK_GROUP_ILP") X(generic_match, "NLK_GENERIC_MATCH") X(mtx, "NLK_MTX") X(mty, "NLK_MTY")    \
  X(match_wg8, "NLK_MATCH_WG8") X(match_bx2, "NLK_MATCH_BX2") X(match_block, "NLK_MATCH_BLOCK")              \
  X(match_noblock, "NLK_MATCH_NOBLOCK") X(match_order, "NLK_MATCH_ORDER") X(commit_wave, "NLK_COMMIT_WAVE") X(commit_lds, "NLK_COMMIT_LDS")    \
  X(commit_band, "NLK_COMMIT_BAND") X(no_chase, "NLK_NO_CHASE") X(chase_test_skip0, "NLK_CHASE_TEST_SKIP0") X(bands, "NLK_BANDS") X(host_bands, "NLK_HOST_BANDS")                    \
  X(host_trace, "NLK_HOST_TRACE") X(gtx, "NLK_GTX") X(gty, "NLK_GTY") X(g8_tail, "NLK_G8_TAIL")              \
  X(g8_single, "NLK_G8_SINGLE") X(tv_wg_pixels, "NLK_TV_WG_PIXELS") X(tv_unblocked, "NLK_TV_UNBLOCKED")      \
  X(tv_batch, "NLK_TV_BATCH") X(tv_mid, "NLK_TV_MID") X(tv_shape, "NLK_TV_SHAPE") X(tv_deep, "NLK_TV_DEEP")  \
  X(tv_inline, "NLK_TV_INLINE") X(tv_wg_full, "NLK_TV_WG_FULL") X(tv_look, "NLK_TV_LOOK") X(tv_look2, "NLK_TV_LOOK2") X(tv_trace, "NLK_TV_TRACE")
struct NlkSwitches {
#define NLK_X(field, name) int field = NLK_UNSET;
  NLK_SWITCH_LIST(NLK_X)
#undef NLK_X
  void load() {
#define NLK_X(field, name) { const char* e_ = getenv(name); field = e_ ? atoi(e_) : NLK_UNSET; }
    NLK_SWITCH_LIST(NLK_X)
#undef NLK_X
    // NLK_MATCH_ORDER = exact | block (or 0 | 1): the summation order of the patch distances (k_match.h)
    if (const char* e_ = getenv("NLK_MATCH_ORDER")) match_order = (!strcmp(e_, "block") || atoi(e_) == 1) ? 1 : 0;
  }
};
static inline bool nlk_set(int v) { return v != NLK_UNSET; }                 // the variable exists (any value)
static inline int nlk_or(int v, int dflt) { return v != NLK_UNSET ? v : dflt; }  // its value, or the default

struct nlk_ctx {
  int device = 0;
  NlkSwitches sw;                    // NLK_* switches as of nlk_ctx_create / nlk_ctx_reload_switches
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr;  // second stream of the banded frame pipeline (run_frame)
  hipEvent_t sync_ev[8] = {};        // cross-stream dependencies of that pipeline (no timing)
  NlkRecView rv;                     // set by the orchestration before every launcher call
  bool chase_on = false;             // the group launch being set up replays the mask itself ...
  int chase_row0 = 0, chase_rows = 0;  // ... over the grid rows [0, chase_rows); the launch's first row is chase_row0
  int lazy_ngx = 0, lazy_ngy = 0;    // != 0: the decision bytes of the grid rows [0, lazy_ngy) are still to be expanded
  unsigned char* lazy_dst = nullptr; //        from the tagged words into this array (nlk_ctx_flush_active; read_records)
  uint64_t* marks_ext = nullptr;     // strip calls: the matcher writes its mark words straight into the caller's array
  float* strip_acc = nullptr;        // strip calls: accumulator whose rows the layout kernel clears as it lays them out
  char err[512] = "";
  NlkBuf planes;                  // the planar copies of cur | prev | basic, one allocation (32-bit offsets between them: k_group8m.h)
  NlkBuf rowok, vmap, topk, tinfo, gcoords, marks, active, acc, tabs, wide;
  NlkBuf skew;                    // mark words in replay-step order (k_marks_skew)
  NlkBuf chase;                   // in-launch mask replay: [0] the generation counter, from word 64 on the tagged decision
                                  // words (zeroed when allocated)
  NlkBuf ms;                      // whole-image DCT: temporary image + the two basis matrices
  NlkBuf tv;                      // TV-L1 pyramids and work images
  NlkBuf slab, tflag;             // deterministic aggregation: per-tile accumulator slabs + "written" flags (k_gather.h)
  // host-pointer frame calls (nlk_frame_host): device copies of the caller's images, the streams the row bands
  // travel on and the events that order them against the kernels
  NlkBuf hw_cur, hw_prev, hw_basic, hw_out;
  hipStream_t up_stream = nullptr, dn_stream = nullptr;
  hipEvent_t band_ev[5][8] = {};  // per band: uploaded / laid out / mask rows replayed / groups filtered / rows normalised
  bool deterministic = false;     // nlk_ctx_set_deterministic / NLK_DETERMINISTIC=1
  NlkTvMail* tv_host = nullptr;   // pinned: the solver state, posted by the kernels (k_tvl1.h)
  unsigned tv_seq = 0;
  int tabs_psz = 0;
  NlkGeom last{};
  bool have_last = false;
  const float *p_match = nullptr, *p_cur = nullptr, *p_prev = nullptr;  // planar images of the last match phase
  const float* p_diff = nullptr;   // ... and of a smoother call: planar prev - cur (k_layout), else nullptr
  bool layout_diff = false;        // the call being laid out keeps that image
  // profiling: one set of NEV events per frame call, read back (and averaged)
  // only by nlk_ctx_get_timings, so the timed loop never synchronises
  static constexpr int NEV = 7, MAXSETS = 512;
  bool profiling = false;
  bool recording = false;    // the current frame call has an event set (false once MAXSETS are used)
  bool set_open = false;     // between event 0 and event 6 of a call
  unsigned set_seen = 0;     // events of the open set recorded so far
  hipEvent_t* ev = nullptr;  // [MAXSETS][NEV], created lazily
  int nsets = 0;             // completed + current
  nlk_timings tm{};
};

extern char nlk_g_err[512];  // last error without a context (nlk_hip.hip)

static inline int fail(nlk_ctx* c, int code, const char* fmt, ...) {
  char msg[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof msg, fmt, ap);
  va_end(ap);
  snprintf(nlk_g_err, sizeof nlk_g_err, "%s", msg);
  if (c) snprintf(c->err, sizeof c->err, "%s", msg);
  return code;
}

#define HIPCHK(ctx, call)                                                         \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(ctx, NLK_EHIP, "%s failed: %s (%s:%d)", #call,                  \
                  hipGetErrorString(e_), __FILE__, __LINE__);                     \
  } while (0)

// Every entry point that launches, allocates, copies or synchronises makes its context's device the
// calling thread's current one first: several contexts (NLK_DEVICES, host/multidev.c) may be driven
// from one thread in any order, and HIP launches / allocates on the CURRENT device, whatever the stream.
#define NLK_USE_DEVICE(ctx) HIPCHK(ctx, hipSetDevice((ctx)->device))

static inline int reserve(nlk_ctx* c, NlkBuf& b, size_t bytes) {
  if (bytes <= b.cap) return NLK_OK;
  if (b.p) HIPCHK(c, hipFree(b.base ? b.base : b.p));
  b.p = b.base = nullptr;
  b.cap = 0;
  // NLK_DEBUG_GUARD=1 (debugging aid): no slack, and the buffer ends where its allocation ends (allocations are
  // mapped in 2 MiB pieces), so that a kernel reading or writing past the end of a scratch buffer faults instead
  // of landing in the slack of this or in the next allocation
  static const bool guard = getenv("NLK_DEBUG_GUARD") != nullptr;
  if (guard) {
    const size_t page = (size_t)2 << 20, used = (bytes + 255) & ~(size_t)255, want = (used + page - 1) / page * page;
    if (hipMalloc(&b.base, want) != hipSuccess) return fail(c, NLK_ENOMEM, "hipMalloc of %zu bytes failed", want);
    b.p = (char*)b.base + (want - used);
    b.cap = bytes;
    return NLK_OK;
  }
  const size_t want = bytes + bytes / 8 + 256;
  if (hipMalloc(&b.p, want) != hipSuccess)
    return fail(c, NLK_ENOMEM, "hipMalloc of %zu bytes failed", want);
  b.cap = want;
  return NLK_OK;
}

// ---- launchers (one translation unit each; all enqueue on c->stream)
// tu_group8.hip: the matrix-core / register group kernels for 8x8 patches
int nlk_launch_group8(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                      float* acc, const uint8_t* active);
// tu_groupp_{a,b,c}.hip: the packed-lane kernel (k_groupp.h), patch sizes 2..8 / 9..12 / 13..16, any channel count
int nlk_launch_groupp_a(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                        float* acc, const uint8_t* active);
int nlk_launch_groupp_b(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                        float* acc, const uint8_t* active);
int nlk_launch_groupp_c(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                        float* acc, const uint8_t* active);
// tu_group_generic.hip: the LDS-DCT kernel (even patch sizes, candidate lists of any length)
int nlk_launch_group_generic(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                             const float* prev, float* acc, const uint8_t* active);
// ... and the kernel for patch sizes 17..32 (k_group_any.h)
int nlk_launch_group_any(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                         float* acc, const uint8_t* active);
// tu_match.hip: block matching + selection (wide = the queued targets of a temporal frame)
int nlk_launch_match(nlk_ctx* c, const NlkGeom& g, const NlkTile& tl, size_t lds, const float* img,
                     int maxm, bool wide);
int nlk_launch_match_generic(nlk_ctx* c, const NlkGeom& g, const float* img);
