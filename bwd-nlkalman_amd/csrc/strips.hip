// strips.hip — one frame over several GPUs, driven from C: row strips of the patch grid, the previous frame's
// halo rows and the accumulator halos between neighbours, the mark words of every strip to every strip
// (nlk_strips_* of include/nlk_hip.h). Reference analogue: the static row split of the OpenMP loop,
// src/nlkalman.c:586 - with the processed-mask replayed over the WHOLE grid on every rank, so that the result
// is the serial order's for any number of strips.
//
// Two transports behind one step:
//   * RCCL (one process per GPU, one strip per process): grouped ncclSend / ncclRecv between neighbours over
//     xGMI, and the mark words as one group of ncclBroadcast - every rank's rows land at their place in the
//     whole-grid array (an all-gather with ragged counts, no padding, no compaction). librccl is dlopen'ed:
//     the library the process already holds (torch's) if there is one, so that the product itself has no link
//     dependency on it.
//   * device copies (every strip of the frame in ONE process: the listed devices, which may repeat - how a
//     one-GPU box runs the N-strip decomposition): hipMemcpyPeerAsync ordered by events.
// A step only enqueues: no allocation, no host synchronisation, one pass over fixed buffers - which is also
// what lets the whole step be captured into a HIP graph and replayed (nlk_strips_set_graph).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <time.h>

#include <vector>

#include "nlk_internal.h"

namespace {

struct StripPlan {
  int gy0, gy1, Y0, Y1, own0, own1;
};

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
char g_rccl_path[256] = "";

// the RCCL the process already has (torch.distributed loads its own copy), else the ROCm one
int load_rccl() {
  if (g_rccl.lib) return NLK_OK;
  const char* names[] = {getenv("NLK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (int pass = 0; pass < 2 && !lib; ++pass)        // pass 0: only what is loaded already
    for (const char* n : names) {
      if (!n) continue;
      lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (lib) { snprintf(g_rccl_path, sizeof g_rccl_path, "%s%s", n, pass == 0 ? " (already loaded)" : ""); break; }
    }
  if (!lib) return fail(nullptr, NLK_EHIP, "cannot load librccl (%s)", dlerror());
#define SYM(field, name)                                                                \
  *(void**)&g_rccl.field = dlsym(lib, name);                                            \
  if (!g_rccl.field) return fail(nullptr, NLK_EHIP, "librccl has no symbol %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(Broadcast, "ncclBroadcast"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.lib = lib;
  return NLK_OK;
}

#define NCCLCHK(ctx, call)                                                                               \
  do {                                                                                                   \
    ncclResult_t r_ = (call);                                                                            \
    if (r_ != ncclSuccess)                                                                               \
      return fail(ctx, NLK_EHIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// rows [r0, r0 + nr) of every plane of a planar (np, hl, w) accumulator <-> a packed (np, nr, w) buffer; the rows
// above the strip's own rows (a) and below them (b) in ONE launch each way: blockIdx.y < na is side a
struct RowSide { float* buf; int r0, nr; };
__global__ void k_pack_rows(const float* __restrict__ acc, int w, int hl, RowSide a, RowSide b) {
  const int p = blockIdx.z;
  const bool first = (int)blockIdx.y < a.nr;
  const RowSide s = first ? a : b;
  const int r = first ? blockIdx.y : blockIdx.y - a.nr;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x)
    s.buf[((size_t)p * s.nr + r) * w + x] = acc[((size_t)p * hl + s.r0 + r) * w + x];
}
__global__ void k_add_rows(float* __restrict__ acc, int w, int hl, RowSide a, RowSide b) {
  const int p = blockIdx.z;
  const bool first = (int)blockIdx.y < a.nr;
  const RowSide s = first ? a : b;
  const int r = first ? blockIdx.y : blockIdx.y - a.nr;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x)
    acc[((size_t)p * hl + s.r0 + r) * w + x] += s.buf[((size_t)p * s.nr + r) * w + x];
}

enum { PH_PREV, PH_MATCH, PH_MARKS, PH_COMMIT, PH_GROUP, PH_ACC, PH_NORM, PH_N };

struct Strip {
  nlk_ctx* c = nullptr;
  int rank = 0, device = 0;
  StripPlan p{};
  int hl = 0, oy = 0, rows = 0;                       // strip height (pixels), first target row, target rows
  int n_up = 0, n_dn = 0, h_top = 0, h_bot = 0;       // rows the neighbours read of mine / my halo rows
  int i0 = 0, i1 = 0;                                 // target rows that see only my own rows of the previous frame
  float *cur = nullptr, *prev = nullptr, *out = nullptr, *acc = nullptr;
  float *snd_top = nullptr, *snd_bot = nullptr, *rcv_top = nullptr, *rcv_bot = nullptr;
  unsigned long long* marks_full = nullptr;
  unsigned char* active_full = nullptr;
  bool chased = false;  // the last step's group kernel replayed the mask itself (active_full: on request, own_rows)
  hipStream_t comm = nullptr;                         // the halo exchange runs beside the interior's matching
  hipEvent_t ev_start = nullptr, ev_prev = nullptr, ev_match = nullptr, ev_group = nullptr, ev_packed = nullptr;
  hipEvent_t ph[PH_N + 1] = {};                       // phase boundaries on the main stream (timing)
  hipGraphExec_t graph = nullptr;
};

}  // namespace

struct nlk_strips {
  int world = 1, nlocal = 1, rank0 = 0;
  int w = 0, h = 0, ch = 0, psz = 0, step = 0, halo = 0, ngx = 0, ngy = 0, smoother = 0, reach = 0;
  float sigma = 0;
  nlkalman_params P{};
  std::vector<StripPlan> plan;       // every rank of the world
  std::vector<Strip> s;              // the local ones
  ncclComm_t comm = nullptr;
  bool rccl = false, overlap = false, have_prev = true, timing = false, want_graph = false, graph_failed = false;
  bool dry = false;  // nlk_strips_set_dry_run: exchanges skipped (one rank of a larger world timed alone)
  int timed_steps = 0, steps_done = 0;
  double phase_ms[PH_N] = {}, issue_us = 0;
  int issue_n = 0;
  char err[512] = "";
};

namespace {

int sfail(nlk_strips* S, int code, const char* what, nlk_ctx* c) {
  snprintf(S->err, sizeof S->err, "%s: %s", what, nlk_last_error(c));
  snprintf(nlk_g_err, sizeof nlk_g_err, "%s", S->err);
  return code;
}
#define SCHK(S, ctx, call)                                   \
  do {                                                       \
    int rc_ = (call);                                        \
    if (rc_) return sfail(S, rc_, #call, ctx);               \
  } while (0)

Strip* local_of(nlk_strips* S, int rank) {
  return (rank >= S->rank0 && rank < S->rank0 + S->nlocal) ? &S->s[rank - S->rank0] : nullptr;
}

// device-to-device copy on `stream` of strip D (any two devices, or one)
int copy_between(nlk_strips* S, Strip& D, void* dst, const Strip& src_strip, const void* src, size_t bytes, hipStream_t stream) {
  HIPCHK(D.c, hipSetDevice(D.device));
  if (D.device == src_strip.device) HIPCHK(D.c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
  else HIPCHK(D.c, hipMemcpyPeerAsync(dst, D.device, src, src_strip.device, bytes, stream));
  (void)S;
  return NLK_OK;
}

// ---- the phases of one step, for every local strip (phase by phase: a local neighbour's data must be enqueued
// before the copy that reads it)
int enqueue_step(nlk_strips* S) {
  const int w = S->w, ch = S->ch, psz = S->psz, ngx = S->ngx;
  const size_t row = sizeof(float) * (size_t)w * ch;
  const bool rec = S->timing;
  auto tick = [&](Strip& T, int i) -> int {
    if (rec) HIPCHK(T.c, hipEventRecord(T.ph[i], T.c->stream));
    return NLK_OK;
  };
  // (1) halo rows of the previous frame: rows of `prev` are contiguous, sent from and received into place
  for (Strip& T : S->s) {
    HIPCHK(T.c, hipSetDevice(T.device));
    int rc = tick(T, PH_PREV);
    if (rc) return rc;
    HIPCHK(T.c, hipEventRecord(T.ev_start, T.c->stream));
  }
  if (S->have_prev && S->world > 1)
    for (Strip& T : S->s) {
      HIPCHK(T.c, hipSetDevice(T.device));
      HIPCHK(T.c, hipStreamWaitEvent(T.comm, T.ev_start, 0));
      if (S->dry) { HIPCHK(T.c, hipEventRecord(T.ev_prev, T.comm)); continue; }
      const int up = T.rank - 1, dn = T.rank + 1;
      float* top_own = T.prev + (size_t)(T.p.own0 - T.p.Y0) * w * ch;           // my first own rows
      float* bot_own = T.prev + (size_t)(T.p.own1 - T.n_dn - T.p.Y0) * w * ch;  // my last own rows the lower rank reads
      float* bot_halo = T.prev + (size_t)(T.p.own1 - T.p.Y0) * w * ch;
      if (S->rccl) {
        NCCLCHK(T.c, g_rccl.GroupStart());
        if (up >= 0) {
          NCCLCHK(T.c, g_rccl.Send(top_own, (size_t)T.n_up * w * ch, ncclFloat, up, S->comm, T.comm));
          NCCLCHK(T.c, g_rccl.Recv(T.prev, (size_t)T.h_top * w * ch, ncclFloat, up, S->comm, T.comm));
        }
        if (dn < S->world) {
          NCCLCHK(T.c, g_rccl.Send(bot_own, (size_t)T.n_dn * w * ch, ncclFloat, dn, S->comm, T.comm));
          NCCLCHK(T.c, g_rccl.Recv(bot_halo, (size_t)T.h_bot * w * ch, ncclFloat, dn, S->comm, T.comm));
        }
        NCCLCHK(T.c, g_rccl.GroupEnd());
      } else {
        // my halo rows are the neighbour's own rows: pulled from its strip once its stream has reached this step
        if (up >= 0) {
          Strip* U = local_of(S, up);
          HIPCHK(T.c, hipStreamWaitEvent(T.comm, U->ev_start, 0));
          int rc = copy_between(S, T, T.prev, *U, U->prev + (size_t)(T.p.Y0 - U->p.Y0) * w * ch, row * T.h_top, T.comm);
          if (rc) return rc;
        }
        if (dn < S->world) {
          Strip* L = local_of(S, dn);
          HIPCHK(T.c, hipStreamWaitEvent(T.comm, L->ev_start, 0));
          int rc = copy_between(S, T, bot_halo, *L, L->prev + (size_t)(T.p.own1 - L->p.Y0) * w * ch, row * T.h_bot, T.comm);
          if (rc) return rc;
        }
      }
      HIPCHK(T.c, hipEventRecord(T.ev_prev, T.comm));
    }
  // (2) matching: the rows that read only my own rows of the previous frame while the halo travels, the seam
  // rows (and the layout of the halo rows) after its arrival; the mark words land in the whole-grid array
  for (Strip& T : S->s) {
    HIPCHK(T.c, hipSetDevice(T.device));
    int rc = tick(T, PH_MATCH);
    if (rc) return rc;
    unsigned long long* mk = T.marks_full + (size_t)T.p.gy0 * ngx;
    const float* pv = S->have_prev ? T.prev : nullptr;
    int reach = 0;
    const bool halo = S->have_prev && S->world > 1 && (T.rank > 0 || T.rank < S->world - 1);
    // (optional, off by default: the split costs two more rounds of the matching launches - ~70 us at 1080p with
    // 2 or 4 strips, measured on a rank stepped alone - against the < 0.5 MB halo exchange it hides; and only
    // while the interior is most of a sizeable strip)
    if (halo && S->overlap && T.i1 - T.i0 >= 48 && 4 * (T.i1 - T.i0) >= 3 * T.rows) {
      const int o0 = T.p.own0 - T.p.Y0, o1 = T.p.own1 - T.p.Y0, hl = T.hl;
      SCHK(S, T.c, nlk_dev_strip_match_part(T.c, T.cur, pv, nullptr, w, hl, ch, S->sigma, &S->P, T.oy, T.rows, S->smoother,
                                            T.i0, T.i1 - T.i0, o0, o1, o0, o1 == hl ? o1 : o1 - psz + 1, mk, &reach));
      HIPCHK(T.c, hipStreamWaitEvent(T.c->stream, T.ev_prev, 0));
      if (o0 > 0 || T.i0 > 0)
        SCHK(S, T.c, nlk_dev_strip_match_part(T.c, T.cur, pv, nullptr, w, hl, ch, S->sigma, &S->P, T.oy, T.rows, S->smoother,
                                              0, T.i0, 0, o0, 0, o0, mk, &reach));
      if (o1 < hl || T.i1 < T.rows) {
        const int v0 = o1 < hl ? (o1 - psz + 1 > o0 ? o1 - psz + 1 : o0) : hl;
        SCHK(S, T.c, nlk_dev_strip_match_part(T.c, T.cur, pv, nullptr, w, hl, ch, S->sigma, &S->P, T.oy, T.rows, S->smoother,
                                              T.i1, T.rows - T.i1, o1, hl, v0, hl, mk, &reach));
      }
    } else {
      if (halo) HIPCHK(T.c, hipStreamWaitEvent(T.c->stream, T.ev_prev, 0));
      SCHK(S, T.c, nlk_dev_strip_match(T.c, T.cur, pv, nullptr, w, T.hl, ch, S->sigma, &S->P, T.oy, T.rows, S->smoother, mk, &reach));
    }
    S->reach = reach;
    HIPCHK(T.c, hipEventRecord(T.ev_match, T.c->stream));
  }
  // (3) every strip's mark words to every strip
  for (Strip& T : S->s) {
    HIPCHK(T.c, hipSetDevice(T.device));
    int rc = tick(T, PH_MARKS);
    if (rc) return rc;
    if (S->world == 1 || S->dry) continue;
    if (S->rccl) {
      NCCLCHK(T.c, g_rccl.GroupStart());
      for (int r = 0; r < S->world; ++r) {
        unsigned long long* at = T.marks_full + (size_t)S->plan[r].gy0 * ngx;
        NCCLCHK(T.c, g_rccl.Broadcast(at, at, (size_t)(S->plan[r].gy1 - S->plan[r].gy0) * ngx, ncclUint64, r, S->comm, T.c->stream));
      }
      NCCLCHK(T.c, g_rccl.GroupEnd());
    } else {
      for (Strip& O : S->s) {
        if (&O == &T) continue;
        HIPCHK(T.c, hipStreamWaitEvent(T.c->stream, O.ev_match, 0));
        const size_t off = (size_t)O.p.gy0 * ngx;
        int rc2 = copy_between(S, T, T.marks_full + off, O, O.marks_full + off, sizeof(unsigned long long) * (size_t)O.rows * ngx, T.c->stream);
        if (rc2) return rc2;
      }
    }
  }
  // (4) the raster-order mask over the whole grid, (5) the strip's groups
  for (Strip& T : S->s) {
    HIPCHK(T.c, hipSetDevice(T.device));
    int rc = tick(T, PH_COMMIT);
    if (rc) return rc;
    if ((rc = tick(T, PH_GROUP))) return rc;
    // (one call: where the group kernel can, it replays the mask rows down to the strip's last one inside its own
    // launch - the commit phase then reads as zero, its bit-plane kernel is counted with the groups)
    SCHK(S, T.c, nlk_dev_strip_commit_group(T.c, T.acc, T.marks_full, ngx, S->ngy, S->reach, T.p.gy0, T.active_full));
    T.chased = T.c->lazy_ngx != 0;
    if ((rc = tick(T, PH_ACC))) return rc;
    // the accumulator rows written outside my own rows, packed for their owners
    if (T.h_top + T.h_bot > 0) {
      const RowSide a{T.snd_top, 0, T.h_top}, b{T.snd_bot, T.p.own1 - T.p.Y0, T.h_bot};
      hipLaunchKernelGGL(k_pack_rows, dim3((w + 255) / 256, T.h_top + T.h_bot, ch + 1), dim3(256), 0, T.c->stream, T.acc, w, T.hl, a, b);
      HIPCHK(T.c, hipGetLastError());
    }
    HIPCHK(T.c, hipEventRecord(T.ev_packed, T.c->stream));
  }
  // (6) accumulator halos to their owners, added on arrival; (7) own rows normalised
  for (Strip& T : S->s) {
    HIPCHK(T.c, hipSetDevice(T.device));
    const int up = T.rank - 1, dn = T.rank + 1;
    const size_t plane = (size_t)w * (ch + 1);
    if (S->world > 1) {
      if (S->dry) {
      } else if (S->rccl) {
        NCCLCHK(T.c, g_rccl.GroupStart());
        if (up >= 0) {
          NCCLCHK(T.c, g_rccl.Send(T.snd_top, plane * T.h_top, ncclFloat, up, S->comm, T.c->stream));
          NCCLCHK(T.c, g_rccl.Recv(T.rcv_top, plane * T.n_up, ncclFloat, up, S->comm, T.c->stream));
        }
        if (dn < S->world) {
          NCCLCHK(T.c, g_rccl.Send(T.snd_bot, plane * T.h_bot, ncclFloat, dn, S->comm, T.c->stream));
          NCCLCHK(T.c, g_rccl.Recv(T.rcv_bot, plane * T.n_dn, ncclFloat, dn, S->comm, T.c->stream));
        }
        NCCLCHK(T.c, g_rccl.GroupEnd());
      } else {
        if (up >= 0) {  // what the upper strip wrote below its own rows = its snd_bot = my first own rows
          Strip* U = local_of(S, up);
          HIPCHK(T.c, hipStreamWaitEvent(T.c->stream, U->ev_packed, 0));
          int rc = copy_between(S, T, T.rcv_top, *U, U->snd_bot, sizeof(float) * plane * T.n_up, T.c->stream);
          if (rc) return rc;
        }
        if (dn < S->world) {
          Strip* L = local_of(S, dn);
          HIPCHK(T.c, hipStreamWaitEvent(T.c->stream, L->ev_packed, 0));
          int rc = copy_between(S, T, T.rcv_bot, *L, L->snd_top, sizeof(float) * plane * T.n_dn, T.c->stream);
          if (rc) return rc;
        }
      }
      if (T.n_up + T.n_dn > 0) {
        const RowSide a{T.rcv_top, T.p.own0 - T.p.Y0, T.n_up}, b{T.rcv_bot, T.p.own1 - T.n_dn - T.p.Y0, T.n_dn};
        hipLaunchKernelGGL(k_add_rows, dim3((w + 255) / 256, T.n_up + T.n_dn, ch + 1), dim3(256), 0, T.c->stream, T.acc, w, T.hl, a, b);
        HIPCHK(T.c, hipGetLastError());
      }
    }
    int rc = tick(T, PH_NORM);
    if (rc) return rc;
    SCHK(S, T.c, nlk_dev_frame_normalize(T.c, T.out, T.acc, T.cur, w, T.hl, ch, T.p.own0 - T.p.Y0, T.p.own1 - T.p.Y0));
    if ((rc = tick(T, PH_N))) return rc;
    HIPCHK(T.c, hipEventRecord(T.ev_group, T.c->stream));
  }
  // (nobody may overwrite a buffer another local strip still reads: the next step's first writes - the halo
  // rows of `prev`, the packed accumulator rows, and the mark words of the strip's own rows, which EVERY other
  // strip copies on its own stream (not only the neighbours: ADVICE r4) - wait for every local strip's step to be
  // over; ev_group is recorded behind all of a strip's copies)
  if (!S->rccl && !S->dry && S->world > 1)
    for (Strip& T : S->s) {
      HIPCHK(T.c, hipSetDevice(T.device));
      for (Strip& N : S->s)
        if (&N != &T) HIPCHK(T.c, hipStreamWaitEvent(T.c->stream, N.ev_group, 0));
    }
  return NLK_OK;
}

}  // namespace

extern "C" {

const char* nlk_strips_last_error(const nlk_strips* S) { return S ? S->err : nlk_g_err; }

int nlk_rccl_unique_id(void* id128) {
  if (!id128) return fail(nullptr, NLK_EINVAL, "null id buffer");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  NCCLCHK(nullptr, g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return NLK_OK;
}

int nlk_strips_create(nlk_strips** out, int nlocal, const int* devices, int rank0, int world, int w, int h, int ch,
                      float sigma, const struct nlkalman_params* P, int smoother, int have_prev) {
  if (!out || !devices || !P) return fail(nullptr, NLK_EINVAL, "null argument");
  *out = nullptr;
  if (world < 1 || nlocal < 1 || rank0 < 0 || rank0 + nlocal > world || !(nlocal == 1 || nlocal == world))
    return fail(nullptr, NLK_EINVAL, "strips: %d local of %d from rank %d (one strip per process, or all of them in one)", nlocal, world, rank0);
  const int psz = P->patch_sz, step = psz / 2;
  if (psz < 2 || w < psz || h < psz) return fail(nullptr, NLK_EINVAL, "strips: patch size %d on a %dx%d frame", psz, w, h);
  nlk_strips* S = new nlk_strips();
  S->world = world; S->nlocal = nlocal; S->rank0 = rank0;
  S->w = w; S->h = h; S->ch = ch; S->psz = psz; S->step = step; S->sigma = sigma; S->P = *P; S->smoother = smoother;
  S->have_prev = have_prev != 0;
  S->halo = smoother ? P->search_sz_t : (P->search_sz_x > P->search_sz_t ? P->search_sz_x : P->search_sz_t);
  S->ngx = (w - psz) / step + 1;
  S->ngy = (h - psz) / step + 1;
  const int reach = ((smoother || have_prev) ? P->search_sz_t : P->search_sz_x) / step;
  auto bail = [&](int code, const char* msg) { snprintf(nlk_g_err, sizeof nlk_g_err, "%s", msg); nlk_strips_destroy(S); return code; };
  if (reach > 3) return bail(NLK_EUNSUP, "strips: a group reaches more than 3 grid cells (64-bit mark words cannot describe it)");
  if (S->ngy < world) return bail(NLK_EINVAL, "strips: fewer patch-grid rows than ranks");
  // the plan of bwd-nlkalman_amd/strips.py: rows of the patch grid, pixel rows incl. the search halo, own rows
  S->plan.resize(world);
  for (int r = 0; r < world; ++r) {
    StripPlan& q = S->plan[r];
    q.gy0 = (int)((long)S->ngy * r / world);
    q.gy1 = (int)((long)S->ngy * (r + 1) / world);
    q.Y0 = q.gy0 * step - S->halo > 0 ? q.gy0 * step - S->halo : 0;
    q.Y1 = (q.gy1 - 1) * step + S->halo + psz < h ? (q.gy1 - 1) * step + S->halo + psz : h;
    q.own0 = r > 0 ? q.gy0 * step : 0;
    q.own1 = r < world - 1 ? q.gy1 * step : h;
  }
  for (int r = 0; r + 1 < world; ++r)
    if (S->plan[r].Y1 > S->plan[r + 1].own1 || S->plan[r + 1].Y0 < S->plan[r].own0)
      return bail(NLK_EINVAL, "strips thinner than the search halo: use fewer ranks");
  S->s.resize(nlocal);
  for (int i = 0; i < nlocal; ++i) {
    Strip& T = S->s[i];
    T.rank = rank0 + i;
    T.device = devices[i];
    T.p = S->plan[T.rank];
    int rc = nlk_ctx_create(&T.c, T.device);
    if (rc) { nlk_strips_destroy(S); return rc; }
    T.hl = T.p.Y1 - T.p.Y0;
    T.oy = T.p.gy0 * step - T.p.Y0;
    T.rows = T.p.gy1 - T.p.gy0;
    const int up = T.rank - 1, dn = T.rank + 1;
    T.n_up = up >= 0 ? S->plan[up].Y1 - T.p.own0 : 0;
    T.n_dn = dn < world ? T.p.own1 - S->plan[dn].Y0 : 0;
    T.h_top = T.p.own0 - T.p.Y0;
    T.h_bot = T.p.Y1 - T.p.own1;
    // target rows whose windows and candidate patches lie inside my own rows of the previous frame
    int lo = up < 0 ? T.p.gy0 : (T.p.own0 + S->halo + step - 1) / step;
    int hi = dn >= world ? T.p.gy1 : (T.p.own1 - S->halo - psz) / step + 1;
    lo = lo < T.p.gy0 ? T.p.gy0 : lo;
    hi = hi > T.p.gy1 ? T.p.gy1 : hi;
    T.i0 = lo - T.p.gy0;
    T.i1 = (hi > lo ? hi : lo) - T.p.gy0;
    const size_t img = sizeof(float) * (size_t)T.hl * w * ch, plane = sizeof(float) * (size_t)w * (ch + 1);
    void** bufs[] = {(void**)&T.cur, (void**)&T.prev, (void**)&T.out, (void**)&T.acc, (void**)&T.marks_full, (void**)&T.active_full,
                     (void**)&T.snd_top, (void**)&T.snd_bot, (void**)&T.rcv_top, (void**)&T.rcv_bot};
    const size_t sizes[] = {img, img, img, plane * T.hl, sizeof(unsigned long long) * (size_t)S->ngx * S->ngy, (size_t)S->ngx * S->ngy,
                            plane * (T.h_top > 0 ? T.h_top : 1), plane * (T.h_bot > 0 ? T.h_bot : 1),
                            plane * (T.n_up > 0 ? T.n_up : 1), plane * (T.n_dn > 0 ? T.n_dn : 1)};
    for (int b = 0; b < 10; ++b) {
      if ((rc = nlk_dev_alloc(T.c, bufs[b], sizes[b])) || (rc = nlk_dev_zero(T.c, *bufs[b], sizes[b]))) { nlk_strips_destroy(S); return rc; }
    }
    T.comm = T.c->aux_stream;
    (void)nlk_ctx_set_strip_accumulator(T.c, T.acc);   // (cleared row by row by the layout kernel of every step)
    bool ok = hipSetDevice(T.device) == hipSuccess;
    for (hipEvent_t* e : {&T.ev_start, &T.ev_prev, &T.ev_match, &T.ev_group, &T.ev_packed})
      ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t& e : T.ph) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok || nlk_sync(T.c)) { nlk_strips_destroy(S); return fail(nullptr, NLK_EHIP, "strips: events on device %d", T.device); }
  }
  *out = S;
  return NLK_OK;
}

void nlk_strips_destroy(nlk_strips* S) {
  if (!S) return;
  for (Strip& T : S->s) {
    if (!T.c) continue;
    (void)hipSetDevice(T.device);
    (void)nlk_sync(T.c);
    (void)hipStreamSynchronize(T.c->aux_stream);
    if (T.graph) (void)hipGraphExecDestroy(T.graph);
    for (void* b : {(void*)T.cur, (void*)T.prev, (void*)T.out, (void*)T.acc, (void*)T.marks_full, (void*)T.active_full,
                    (void*)T.snd_top, (void*)T.snd_bot, (void*)T.rcv_top, (void*)T.rcv_bot})
      if (b) (void)nlk_dev_free(T.c, b);
    for (hipEvent_t e : {T.ev_start, T.ev_prev, T.ev_match, T.ev_group, T.ev_packed})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : T.ph)
      if (e) (void)hipEventDestroy(e);
  }
  if (S->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(S->comm);
  for (Strip& T : S->s)
    if (T.c) nlk_ctx_destroy(T.c);
  delete S;
}

// one strip per process: the communicator of the world (id128 from nlk_rccl_unique_id on one rank, carried to
// the others by whatever started them)
int nlk_strips_rccl_init(nlk_strips* S, const void* id128) {
  if (!S || !id128) return fail(nullptr, NLK_EINVAL, "null argument");
  if (S->nlocal != 1) return sfail(S, NLK_EINVAL, "RCCL transport: one strip per process", nullptr);
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  HIPCHK(S->s[0].c, hipSetDevice(S->s[0].device));
  NCCLCHK(S->s[0].c, g_rccl.CommInitRank(&S->comm, S->world, id, S->rank0));
  S->rccl = true;
  return NLK_OK;
}

const char* nlk_strips_transport(const nlk_strips* S) {
  static char buf[320];
  if (!S) return "";
  if (S->rccl) snprintf(buf, sizeof buf, "rccl send/recv + broadcast group, %s", g_rccl_path);
  else snprintf(buf, sizeof buf, S->world > 1 ? "device copies (every strip in one process)" : "none (one strip)");
  return buf;
}

// rows of full device-resident frames -> the strip's buffers (cur: strip + halo; prev: OWN rows only, the halo
// rows arrive by exchange in every step). `local` = index among this process's strips; the frames live on that
// strip's device.
int nlk_strips_load(nlk_strips* S, int local, const float* cur_full, const float* prev_full) {
  if (!S || local < 0 || local >= S->nlocal || !cur_full) return fail(nullptr, NLK_EINVAL, "strips_load: bad argument");
  Strip& T = S->s[local];
  const size_t row = sizeof(float) * (size_t)S->w * S->ch;
  SCHK(S, T.c, nlk_d2d(T.c, T.cur, cur_full + (size_t)T.p.Y0 * S->w * S->ch, row * T.hl));
  SCHK(S, T.c, nlk_dev_zero(T.c, T.prev, row * T.hl));
  if (prev_full)
    SCHK(S, T.c, nlk_d2d(T.c, T.prev + (size_t)(T.p.own0 - T.p.Y0) * S->w * S->ch, prev_full + (size_t)T.p.own0 * S->w * S->ch,
                         row * (T.p.own1 - T.p.own0)));
  return nlk_sync(T.c);
}

// One rank of a larger world run ALONE, every exchange skipped (its halo rows and the other strips' mark words are
// whatever the buffers hold: the output means nothing): what a rank's step costs in kernels and launch gaps at
// that world size, measurable on a one-GPU box. bench.py --strip-model.
int nlk_strips_set_dry_run(nlk_strips* S, int on) {
  if (!S) return NLK_EINVAL;
  S->dry = on != 0;
  return NLK_OK;
}

int nlk_strips_set_options(nlk_strips* S, int overlap, int timing, int graph) {
  if (!S) return NLK_EINVAL;
  S->overlap = overlap != 0;
  if ((timing != 0) != S->timing) {
    S->timed_steps = 0;
    for (double& v : S->phase_ms) v = 0;
  }
  S->timing = timing != 0;
  S->want_graph = graph != 0;
  return NLK_OK;
}

// One frame step on every local strip. Asynchronous: returns when everything is enqueued.
int nlk_strips_step(nlk_strips* S) {
  if (!S) return NLK_EINVAL;
  if (S->world > 1 && S->nlocal == 1 && !S->rccl && !S->dry)
    return sfail(S, NLK_EINVAL, "one strip of several in this process: nlk_strips_rccl_init first", nullptr);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int rc = NLK_OK;
  Strip& T0 = S->s[0];
  // (the first steps size the contexts' scratch buffers and upload the tables - allocation and host copies, which
  // a capture does not take: they run as plain launches)
  const bool graphable = S->want_graph && !S->graph_failed && !S->timing && S->nlocal == 1 && S->steps_done >= 2;
  S->steps_done++;
  if (graphable && T0.graph) {
    HIPCHK(T0.c, hipSetDevice(T0.device));
    HIPCHK(T0.c, hipGraphLaunch(T0.graph, T0.c->stream));
  } else if (graphable) {
    // capture the step once (everything it enqueues, the second stream included, joins the capture through
    // the events), then replay it: one launch per step instead of ~25
    HIPCHK(T0.c, hipSetDevice(T0.device));
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(T0.c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      rc = enqueue_step(S);  // (the second stream forks from the first at ev_start and joins it again at ev_prev)
      const hipError_t e = hipStreamEndCapture(T0.c->stream, &g);
      if (rc || e != hipSuccess || !g || hipGraphInstantiate(&T0.graph, g, nullptr, nullptr, 0) != hipSuccess) {
        S->graph_failed = true;
        T0.graph = nullptr;
        (void)hipGetLastError();
        rc = enqueue_step(S);          // (eager from here on)
      } else {
        HIPCHK(T0.c, hipGraphLaunch(T0.graph, T0.c->stream));
      }
      if (g) (void)hipGraphDestroy(g);
    } else {
      S->graph_failed = true;
      (void)hipGetLastError();
      rc = enqueue_step(S);
    }
  } else {
    rc = enqueue_step(S);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  S->issue_us += (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
  S->issue_n++;
  if (!rc && S->timing) {
    // (diagnosis: the phase events are read back per step - this synchronises)
    for (Strip& T : S->s) {
      HIPCHK(T.c, hipSetDevice(T.device));
      HIPCHK(T.c, hipStreamSynchronize(T.c->stream));
    }
    Strip& T = S->s[0];
    for (int i = 0; i < PH_N; ++i) {
      float ms = 0.f;
      HIPCHK(T.c, hipEventElapsedTime(&ms, T.ph[i], T.ph[i + 1]));
      S->phase_ms[i] += ms;
    }
    S->timed_steps++;
  }
  return rc;
}

int nlk_strips_sync(nlk_strips* S) {
  if (!S) return NLK_EINVAL;
  for (Strip& T : S->s) {
    HIPCHK(T.c, hipSetDevice(T.device));
    HIPCHK(T.c, hipStreamSynchronize(T.c->stream));
    HIPCHK(T.c, hipStreamSynchronize(T.comm));
  }
  return NLK_OK;
}

// own rows [*y0, *y1) of the frame: *rows points at row *y0 (device memory of that strip's device, valid until
// the next step)
int nlk_strips_own_rows(nlk_strips* S, int local, int* y0, int* y1, float** rows, void** marks_full, unsigned char** active_full) {
  if (!S || local < 0 || local >= S->nlocal) return fail(nullptr, NLK_EINVAL, "strips_own_rows: bad argument");
  Strip& T = S->s[local];
  if (y0) *y0 = T.p.own0;
  if (y1) *y1 = T.p.own1;
  if (rows) *rows = T.out + (size_t)(T.p.own0 - T.p.Y0) * S->w * S->ch;
  if (marks_full) *marks_full = T.marks_full;
  if (active_full) {
    // (a strip whose group kernel replays the mask itself holds the decisions as tagged words: the bytes of the grid
    // rows [0, its last row) are written now - the whole grid for the last strip; a replayed HIP graph never came
    // through the call that would have noted it, hence from the strip's own flag)
    if (T.chased) {
      T.c->lazy_ngx = S->ngx; T.c->lazy_ngy = T.p.gy0 + T.rows; T.c->lazy_dst = T.active_full;
      HIPCHK(T.c, hipSetDevice(T.device));
      int rc = nlk_ctx_flush_active(T.c);
      if (rc) return rc;
    }
    *active_full = T.active_full;
  }
  return NLK_OK;
}

nlk_ctx* nlk_strips_ctx(nlk_strips* S, int local) { return (S && local >= 0 && local < S->nlocal) ? S->s[local].c : nullptr; }

// strip geometry of a local strip: gy0, gy1 (grid rows), Y0, Y1 (pixel rows held), own0, own1
int nlk_strips_geometry(nlk_strips* S, int local, int geom[6]) {
  if (!S || !geom || local < 0 || local >= S->nlocal) return fail(nullptr, NLK_EINVAL, "strips_geometry: bad argument");
  const StripPlan& p = S->s[local].p;
  const int v[6] = {p.gy0, p.gy1, p.Y0, p.Y1, p.own0, p.own1};
  memcpy(geom, v, sizeof v);
  return NLK_OK;
}

// mean device time per phase of local strip 0 over the steps made with timing on [prev halo, match, mark words,
// mask replay, groups, accumulator halos, normalise], the mean host time one step took to enqueue (us, over all
// steps so far; reset by reading), and whether the steps ran as a replayed graph
int nlk_strips_stats(nlk_strips* S, float phase_ms[7], float* issue_us, int* graph) {
  if (!S) return NLK_EINVAL;
  if (phase_ms)
    for (int i = 0; i < PH_N; ++i) phase_ms[i] = S->timed_steps ? (float)(S->phase_ms[i] / S->timed_steps) : 0.f;
  if (issue_us) *issue_us = S->issue_n ? (float)(S->issue_us / S->issue_n) : 0.f;
  S->issue_us = 0;
  S->issue_n = 0;
  if (graph) *graph = S->s[0].graph != nullptr;
  return NLK_OK;
}

}  // extern "C"
