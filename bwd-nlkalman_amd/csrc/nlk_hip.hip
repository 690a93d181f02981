// nlk_hip.hip — C-ABI (include/nlk_hip.h) over the gfx950 kernels.
//
// One context per (process, device). All work is enqueued on the context's
// stream; scratch buffers are grown on demand and kept, so a steady-state frame
// call performs no allocation (context/alloc time is on the
// CLI critical path).
#include <math.h>
#include <time.h>

#include <algorithm>
#include <mutex>
#include <vector>

// (the 12-point flow graph of the device code, compiled for the host: nlk_host_tables lets the tests pin it)
namespace nlk_host12 {
#define NLK_HD static inline
#include "k_dct12.h"
#undef NLK_HD
}  // namespace nlk_host12

#include "k_commit.h"
#include "k_frame.h"
#include "k_match.h"   // NlkTile (the kernels themselves are compiled in tu_match.hip)
#include "nlk_internal.h"

char nlk_g_err[512] = "";

namespace {

// live contexts (nlk_ctx_reload_switches(NULL) re-reads the environment for all of them)
std::mutex g_live_mu;
std::vector<nlk_ctx*> g_live;

typedef NlkBuf Buf;

// reference: src/nlkalman.c:365-419 ("gaussian"), float/double mix as written there
void host_window(float* W, int psz) {
  float w1[64];
  const float N2 = ((float)psz - 1.) / 2.;
  for (int n = 0; n < psz; ++n) {
    const float s = .4;
    const float x = ((float)n - N2) / N2 / s;
    w1[n] = exp(-.5 * x * x);
  }
  for (int i = 0; i < psz; ++i)
    for (int j = 0; j < psz; ++j) W[i * psz + j] = w1[i] * w1[j];
}

// orthonormal DCT-II basis = FFTW REDFT10 x the reference's scaling
// (reference: src/nlkalman.c:204-212, 281-298)
void host_basis(float* C, int n) {
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < n; ++j) {
      const double s = (k == 0) ? sqrt(1.0 / n) : sqrt(2.0 / n);
      C[k * n + j] = (float)(s * cos(M_PI * (j + 0.5) * k / n));
    }
}

int upload_tables(nlk_ctx* c, int psz) {
  if (c->tabs_psz == psz) return NLK_OK;
  float host[2 * 64 * 64];
  host_basis(host, psz);
  host_window(host + psz * psz, psz);
  int rc = reserve(c, c->tabs, sizeof(float) * 2 * psz * psz);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->tabs.p, host, sizeof(float) * 2 * psz * psz,
                           hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // host[] is on the stack
  c->tabs_psz = psz;
  return NLK_OK;
}

int launch_group(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur,
                 const float* prev, float* acc, const uint8_t* active) {
  // Default: 8x8 patches with 1 or 3 channels on the matrix cores (k_group8m.h: 1.04 ms at C2 against
  // 1.34 ms for the packed-lane kernel), everything else on the packed-lane kernel (k_groupp.h; its
  // per-lane candidate lists hold up to 128 entries). Comparison variants: NLK_GROUP_PACKED=1 (8x8 on
  // k_groupp), NLK_GROUP_DPP=1 (8x8: registers + DPP), NLK_GROUP12_ROWS=1 (12x12: lane = (channel,
  // row)), NLK_GENERIC_GROUP=1 (LDS-DCT kernel, which also takes the lists of more than 128 entries)
  const bool ch13 = g.ch == 1 || g.ch == 3;
  const bool lists_fit = g.kmax <= 128 && g.gstride <= 128;
  if (g.psz > 16) return nlk_launch_group_any(c, g, img, cur, prev, acc, active);  // (k_group_any.h: 17..32)
  if (nlk_set(c->sw.generic_group) || !lists_fit) return nlk_launch_group_generic(c, g, img, cur, prev, acc, active);
  if (g.psz == 8 && ch13 && !nlk_set(c->sw.group_packed))
    return nlk_launch_group8(c, g, img, cur, prev, acc, active);
  if (g.psz <= 8) return nlk_launch_groupp_a(c, g, img, cur, prev, acc, active);
  if (g.psz <= 12) return nlk_launch_groupp_b(c, g, img, cur, prev, acc, active);
  return nlk_launch_groupp_c(c, g, img, cur, prev, acc, active);
}

int check_images(nlk_ctx* c, const void* out, const void* cur, int w, int h, int ch) {
  if (!c) return fail(nullptr, NLK_EINVAL, "null context");
  if (!out || !cur) return fail(c, NLK_EINVAL, "null image pointer");
  if (w <= 0 || h <= 0 || ch <= 0) return fail(c, NLK_EINVAL, "bad image size %dx%dx%d", w, h, ch);
  if (w > 65535 || h > 65535) return fail(c, NLK_EUNSUP, "image side above 65535");
  return NLK_OK;
}

// event i of the current frame call; event 0 opens a new set
// (once MAXSETS frame calls have been recorded, recording stops: the averages then cover the
// first MAXSETS calls, and no set is ever overwritten or left half-recorded)
// A set opens with event 0 and closes with event 6 (normalisation). A strip whose matching comes in
// several calls (nlk_dev_strip_match_rows) stays in ONE set: events 0 / 1 keep their first recording,
// event 2 its last, so that match_ms spans all the parts (and the halo wait between them).
void mark(nlk_ctx* c, int i) {
  if (!c->profiling) return;
  if (i == 0 && !c->set_open) {
    c->recording = c->nsets < nlk_ctx::MAXSETS;
    if (c->recording) c->nsets++;
    c->set_open = true;
    c->set_seen = 0;
  }
  if (!c->recording || c->nsets < 1) return;
  if (i <= 1 && (c->set_seen >> i) & 1u) return;
  c->set_seen |= 1u << i;
  (void)hipEventRecord(c->ev[(c->nsets - 1) * nlk_ctx::NEV + i], c->stream);
  if (i == 6) c->set_open = false;
}

}  // namespace

extern "C" {

int nlk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* nlk_last_error(const nlk_ctx* ctx) { return ctx ? ctx->err : nlk_g_err; }

int nlk_ctx_create(nlk_ctx** out, int device) {
  if (!out) return fail(nullptr, NLK_EINVAL, "null ctx pointer");
  *out = nullptr;
  int n = 0;
  struct timespec tt0;
  clock_gettime(CLOCK_MONOTONIC, &tt0);
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(nullptr, NLK_ENODEV, "no HIP device visible (the HIP path is mandatory: there is no CPU fallback)");
  if (device < 0 || device >= n)
    return fail(nullptr, NLK_ENODEV, "device %d out of range (%d visible)", device, n);
  nlk_ctx* c = new nlk_ctx();
  c->device = device;
  const bool trace = getenv("NLK_CLI_TRACE") != nullptr;   // (where a one-shot process spends its start-up)
  auto lap = [&](const char* what) {
    if (!trace) return;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[nlk_ctx_create +%7.2f ms] %s\n", (t.tv_sec - tt0.tv_sec) * 1e3 + (t.tv_nsec - tt0.tv_nsec) * 1e-6, what);
  };
  lap("hipGetDeviceCount (runtime initialised)");
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(nullptr, NLK_EHIP, "cannot create a stream on device %d", device);
  }
  lap("first stream");
  c->stream = c->own_stream;
  c->sw.load();
  c->deterministic = nlk_or(c->sw.deterministic, 0) != 0;
  bool ok = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) == hipSuccess;
  for (hipEvent_t& e : c->sync_ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    nlk_ctx_destroy(c);
    return fail(nullptr, NLK_EHIP, "cannot create the second stream / events on device %d", device);
  }
  lap("second stream + events");
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    g_live.push_back(c);
  }
  *out = c;
  return NLK_OK;
}

void nlk_ctx_destroy(nlk_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end());
  }
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->aux_stream) hipStreamSynchronize(c->aux_stream);
  Buf* bufs[] = {&c->planes, &c->rowok, &c->vmap, &c->topk,
                 &c->tinfo, &c->gcoords, &c->marks, &c->active, &c->acc, &c->tabs, &c->wide, &c->tv, &c->skew, &c->chase, &c->ms,
                 &c->slab, &c->tflag, &c->hw_cur, &c->hw_prev, &c->hw_basic, &c->hw_out};
  for (Buf* b : bufs)
    if (b->p) hipFree(b->base ? b->base : b->p);
  if (c->tv_host) (void)hipHostFree(c->tv_host);
  if (c->ev) {
    for (int i = 0; i < nlk_ctx::MAXSETS * nlk_ctx::NEV; ++i) (void)hipEventDestroy(c->ev[i]);
    free(c->ev);
  }
  for (hipEvent_t e : c->sync_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& row : c->band_ev)
    for (hipEvent_t e : row)
      if (e) (void)hipEventDestroy(e);
  if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
  if (c->dn_stream) (void)hipStreamDestroy(c->dn_stream);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int nlk_ctx_set_profiling(nlk_ctx* c, int on) {
  if (!c) return NLK_EINVAL;
  if (on && !c->ev) {
    const int n = nlk_ctx::MAXSETS * nlk_ctx::NEV;
    c->ev = (hipEvent_t*)calloc(n, sizeof(hipEvent_t));
    if (!c->ev) return fail(c, NLK_ENOMEM, "event pool");
    for (int i = 0; i < n; ++i) HIPCHK(c, hipEventCreate(&c->ev[i]));
  }
  c->profiling = on != 0;
  c->recording = false;
  c->set_open = false;
  c->nsets = 0;  // (re)start averaging
  return NLK_OK;
}

// averages over every frame call recorded since profiling was switched on
int nlk_ctx_get_timings(nlk_ctx* c, struct nlk_timings* t) {
  if (!c || !t) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  memset(t, 0, sizeof *t);
  if (!c->ev || c->nsets == 0) return NLK_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const int nclosed = c->nsets - ((c->set_open && c->recording) ? 1 : 0);  // (a call still under way is left out)
  if (nclosed <= 0) return NLK_OK;
  for (int s = 0; s < nclosed; ++s) {
    hipEvent_t* e = c->ev + s * nlk_ctx::NEV;
    const int a[6] = {0, 1, 2, 3, 5, 0}, b[6] = {1, 2, 3, 4, 6, 6};
    for (int i = 0; i < 6; ++i) {
      float ms = 0.f;
      HIPCHK(c, hipEventElapsedTime(&ms, e[a[i]], e[b[i]]));
      acc[i] += ms;
    }
  }
  float* dst[6] = {&t->layout_ms, &t->match_ms, &t->commit_ms, &t->group_ms, &t->normalize_ms,
                   &t->total_ms};
  for (int i = 0; i < 6; ++i) *dst[i] = (float)(acc[i] / nclosed);
  c->tm = *t;
  return NLK_OK;
}

int nlk_ctx_set_deterministic(nlk_ctx* c, int on) {
  if (!c) return NLK_EINVAL;
  c->deterministic = on != 0;
  return NLK_OK;
}

int nlk_ctx_reload_switches(nlk_ctx* c) {
  std::lock_guard<std::mutex> lock(g_live_mu);
  for (nlk_ctx* x : g_live)
    if (!c || x == c) {
      x->sw.load();
      if (nlk_set(x->sw.deterministic)) x->deterministic = x->sw.deterministic != 0;
    }
  return NLK_OK;
}

int nlk_ctx_set_stream(nlk_ctx* c, void* s) {
  if (!c) return NLK_EINVAL;
  c->stream = (hipStream_t)s;  // NULL is the legacy default stream (what torch uses by default)
  return NLK_OK;
}

int nlk_ctx_use_own_stream(nlk_ctx* c) {
  if (!c) return NLK_EINVAL;
  c->stream = c->own_stream;
  return NLK_OK;
}

void* nlk_ctx_get_stream(nlk_ctx* c) { return c ? (void*)c->stream : nullptr; }

int nlk_dev_alloc(nlk_ctx* c, void** d, size_t bytes) {
  if (!c || !d) return fail(c, NLK_EINVAL, "null argument");
  NLK_USE_DEVICE(c);
  if (hipMalloc(d, bytes) != hipSuccess) return fail(c, NLK_ENOMEM, "hipMalloc(%zu) failed", bytes);
  return NLK_OK;
}
int nlk_dev_free(nlk_ctx* c, void* d) {
  if (!c) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipFree(d));
  return NLK_OK;
}
int nlk_h2d(nlk_ctx* c, void* d, const void* h, size_t n) {
  if (!c) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NLK_OK;
}
int nlk_d2h(nlk_ctx* c, void* h, const void* d, size_t n) {
  if (!c) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NLK_OK;
}
int nlk_d2d(nlk_ctx* c, void* dst, const void* src, size_t n) {
  if (!c) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, c->stream));
  return NLK_OK;
}
int nlk_dev_zero(nlk_ctx* c, void* d, size_t n) {
  if (!c || !d) return fail(c, NLK_EINVAL, "null argument");
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipMemsetAsync(d, 0, n, c->stream));
  return NLK_OK;
}
int nlk_dev_add(nlk_ctx* c, float* dst, const float* src, size_t n) {
  if (!c || !dst || !src) return fail(c, NLK_EINVAL, "null argument");
  if (n == 0) return NLK_OK;
  NLK_USE_DEVICE(c);
  hipLaunchKernelGGL(k_add, dim3(1024), dim3(256), 0, c->stream, dst, src, n);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}
// everything enqueued on src_ctx's stream so far has run when the copy starts; the copy itself is
// ordered on dst_ctx's stream
int nlk_dev_copy_peer(nlk_ctx* dc, void* dst, nlk_ctx* sc, const void* src, size_t n) {
  if (!dc || !sc || !dst || !src) return fail(dc, NLK_EINVAL, "null argument");
  HIPCHK(dc, hipSetDevice(sc->device));
  HIPCHK(dc, hipEventRecord(sc->sync_ev[7], sc->stream));
  HIPCHK(dc, hipSetDevice(dc->device));
  HIPCHK(dc, hipStreamWaitEvent(dc->stream, sc->sync_ev[7], 0));
  HIPCHK(dc, hipMemcpyPeerAsync(dst, dc->device, src, sc->device, n, dc->stream));
  return NLK_OK;
}
int nlk_host_alloc(nlk_ctx* c, void** h, size_t n) {
  if (!c || !h) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipHostMalloc(h, n, hipHostMallocDefault));
  return NLK_OK;
}
int nlk_host_free(nlk_ctx* c, void* h) {
  if (!c) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  if (h) HIPCHK(c, hipHostFree(h));
  return NLK_OK;
}
int nlk_sync(nlk_ctx* c) {
  if (!c) return NLK_EINVAL;
  NLK_USE_DEVICE(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NLK_OK;
}

int nlk_dev_rgb2opp(nlk_ctx* c, float* im, int w, int h, int ch) {
  int rc = check_images(c, im, im, w, h, ch);
  if (rc) return rc;
  NLK_USE_DEVICE(c);
  if (ch != 3) return NLK_OK;
  hipLaunchKernelGGL(k_rgb2opp, dim3(2048), dim3(256), 0, c->stream, im, (size_t)w * h);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

int nlk_dev_opp2rgb(nlk_ctx* c, float* im, int w, int h, int ch) {
  int rc = check_images(c, im, im, w, h, ch);
  if (rc) return rc;
  NLK_USE_DEVICE(c);
  if (ch != 3) return NLK_OK;
  hipLaunchKernelGGL(k_opp2rgb, dim3(2048), dim3(256), 0, c->stream, im, (size_t)w * h);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

int nlk_dev_warp_bicubic(nlk_ctx* c, float* imw, const float* im, const float* of,
                         const float* msk, int w, int h, int ch) {
  int rc = check_images(c, imw, im, w, h, ch);
  if (rc) return rc;
  NLK_USE_DEVICE(c);
  if (!of) return fail(c, NLK_EINVAL, "null flow");
  auto kern = ch == 3 ? k_warp_bicubic<3> : (ch == 1 ? k_warp_bicubic<1> : k_warp_bicubic<0>);
  hipLaunchKernelGGL(kern, dim3((w + 127) / 128, h), dim3(128), 0, c->stream, imw,
                     im, of, msk, w, h, ch);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

// ---- frame plan: geometry, planar images, scratch and the launch shapes of one frame or strip call
struct NlkPlan {
  NlkGeom g{};                   // the whole call: target rows [0, g.ngy) from image row g.oy
  const float *img_cur = nullptr, *img_prev = nullptr, *img_basic = nullptr, *img_match = nullptr;
  const float* img_diff = nullptr;   // smoother calls: planar prev - cur
  NlkTile tl{}, tw{};            // match tiles: dominant window / wide window
  size_t lds = 0, lds_w = 0;
  int maxm = 0, maxm_w = 0;
  bool wide = false;
  bool wide0_zeroed = false;     // the layout kernel has cleared the queue length of band 0 (rows from 0)
  bool generic = false;          // k_bm_generic instead of the tiled kernels
};

// records of the target rows from `r0` on (a view into the call's buffers)
static NlkRecView view_rows(nlk_ctx* c, const NlkGeom& g, hipStream_t stream, int r0, int band) {
  const size_t t0 = (size_t)r0 * g.ngx;
  NlkRecView v;
  v.stream = stream;
  v.topk = (uint32_t*)c->topk.p + t0 * g.kmax;
  v.tinfo = (NlkTarget*)c->tinfo.p + t0;
  v.gcoords = (uint32_t*)c->gcoords.p + t0 * g.gstride;
  v.marks = (c->marks_ext ? c->marks_ext : (uint64_t*)c->marks.p) + t0;
  v.wide = (uint32_t*)c->wide.p + t0 + band;  // (every band: its own counter in front of its own list)
  return v;
}

// the geometry of target rows [r0, r0 + rows) of the plan
static NlkGeom band_geom(const NlkGeom& g, int r0, int rows) {
  NlkGeom b = g;
  b.oy = g.oy + r0 * g.step;
  b.ngy = rows;
  return b;
}

// layout of the pixel rows [y0, y1) (planar copies, row test of the validity map, accumulator rows cleared) and
// the validity map's column test for the rows [v0, v1) (a row needs the row tests of the psz rows from it on)
static int layout_rows(nlk_ctx* c, const float* cur, const float* prev, const float* basic, float* acc_zero, int w,
                       int h, int ch, int psz, int y0, int y1, int v0, int v1, uint32_t* zero_word = nullptr) {
  // (every image is copied into the context's slab, one channel too: the group kernel addresses all of them from
  // one base pointer)
  const size_t img_floats = (size_t)w * h * ch;
  float* const slab = (float*)c->planes.p;
  if (y1 > y0)
    hipLaunchKernelGGL(ch == 3 ? k_layout<3> : k_layout<0>, dim3((w + 255) / 256, y1 - y0), dim3(256), 0, c->stream, cur, slab, prev,
                       slab + img_floats, basic, slab + 2 * img_floats, (uint8_t*)c->rowok.p, acc_zero, w, h, ch, psz,
                       1, y0, zero_word, (c->layout_diff && prev) ? slab + 3 * img_floats : nullptr);
  if (prev && v1 > v0) {
    if (w % 4 == 0)
      hipLaunchKernelGGL(k_nan_cols4, dim3((w / 4 + 255) / 256, v1 - v0), dim3(256), 0, c->stream,
                         (const uint32_t*)c->rowok.p, (uint32_t*)c->vmap.p, w / 4, h, psz, v0);
    else
      hipLaunchKernelGGL(k_nan_cols, dim3((w + 255) / 256, v1 - v0), dim3(256), 0, c->stream, (const uint8_t*)c->rowok.p,
                         (uint8_t*)c->vmap.p, w, h, psz, v0);
  }
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

// phase 0: checks, geometry, layout (planar copies + validity map), scratch. Runs on c->stream.
static int plan_frame(nlk_ctx* c, NlkPlan& pl, const float* cur, const float* prev, const float* basic, int w,
                      int h, int ch, float sigma, const struct nlkalman_params* P, int oy, int ngy,
                      int smoother, int nbands, float* acc_zero = nullptr, bool do_layout = true) {
  int rc = check_images(c, cur, cur, w, h, ch);
  if (rc) return rc;
  if (!P) return fail(c, NLK_EINVAL, "null parameters");
  NLK_USE_DEVICE(c);
  c->lazy_ngx = c->lazy_ngy = 0;  // (whatever replays this call's mask: `active` holds its bytes unless set again)
  c->lazy_dst = nullptr;
  NlkGeom& g = pl.g;
  g.w = w; g.h = h; g.ch = ch;
  g.psz = P->patch_sz;
  if (g.psz < 2 || g.psz > 32 || ch * g.psz * g.psz > 4096)
    return fail(c, NLK_EUNSUP, "patch size %d with %d channels not supported (2..32, ch * psz^2 <= 4096)", g.psz, ch);
  g.step = g.psz / 2;
  g.p2 = g.psz * g.psz;
  g.E = g.p2 * ch;
  if (w < g.psz || h < g.psz) return fail(c, NLK_EINVAL, "image smaller than a patch");
  g.ngx = (w - g.psz) / g.step + 1;
  g.oy = oy;
  g.ngy = ngy;
  if (oy < 0 || ngy < 1 || oy + (ngy - 1) * g.step + g.psz > h)
    return fail(c, NLK_EINVAL, "target rows [%d + j*%d, j < %d) leave the %d-row strip", oy,
                g.step, ngy, h);
  g.wsz_x = P->search_sz_x; g.wsz_t = P->search_sz_t;
  g.npx = P->npatches_x; g.npt = P->npatches_t; g.ntagg = P->npatches_tagg;
  if (g.wsz_x < 0 || g.wsz_t < 0 || g.ntagg < 0)
    return fail(c, NLK_EINVAL, "negative search radius or group size (call nlkalman_default_params first)");
  g.have_prev = prev != nullptr;
  g.have_basic = basic != nullptr;
  g.smoother = smoother != 0;
  g.sigma2 = sigma * sigma;
  g.beta_x = P->beta_x; g.beta_t = P->beta_t;
  const int wmax = g.smoother ? g.wsz_t : (g.have_prev ? max(g.wsz_x, g.wsz_t) : g.wsz_x);
  // reach (in grid cells) of the groups that mark the processed-mask: in a
  // temporal frame only groups with a valid previous patch mark (reference: :931)
  // and those searched with the temporal radius (reference: :637)
  const int wmark = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
  g.R = wmark / g.step;  // (R > 3: the mask replay runs from the coordinate lists, commit_rows)
  const int ncand = (2 * wmax + 1) * (2 * wmax + 1);
  // the tiled matching kernels exist for patch sizes 4 / 6 / 8 / 10 / 12 / 16, 1 or 3 channels and
  // windows of up to 1024 candidates; everything else the reference accepts goes to k_bm_generic
  pl.generic = !(g.psz == 4 || g.psz == 6 || g.psz == 8 || g.psz == 10 || g.psz == 12 || g.psz == 16) ||
               !(ch == 1 || ch == 3) || ncand > 64 * 16 || nlk_set(c->sw.generic_match);
  g.kmax = max(max(g.npx, g.npt), 1);
  const int ngrid = g.ngx * g.ngy;
  const int npix = w * h;
  const int ntagg_alloc = max(g.ntagg, 1);
  g.gstride = ntagg_alloc;

  mark(c, 0);
  // ---- layout: planar copies + the row test of the validity map (+ clearing the accumulator of a
  // whole-frame call) in one kernel, the column test in a second
  {
    // planar copies of the (up to) three images in ONE allocation: [cur | prev | basic]
    const size_t img_floats = (size_t)npix * ch;
    // (a smoother call keeps a fourth image, prev - cur: what the pass B of k_group8m<.., true, ..> transforms)
    c->layout_diff = g.smoother && prev != nullptr;
    if ((rc = reserve(c, c->planes, sizeof(float) * img_floats * (c->layout_diff ? 4 : (basic ? 3 : (prev ? 2 : 1)))))) return rc;
    pl.img_diff = c->layout_diff ? (const float*)c->planes.p + 3 * img_floats : nullptr;
    c->p_diff = pl.img_diff;
    pl.img_cur = (const float*)c->planes.p;
    pl.img_prev = prev ? pl.img_cur + img_floats : nullptr;
    pl.img_basic = basic ? pl.img_cur + 2 * img_floats : nullptr;
    if (prev && ((rc = reserve(c, c->rowok, npix)) || (rc = reserve(c, c->vmap, npix)))) return rc;
    if ((rc = reserve(c, c->wide, sizeof(uint32_t) * ((size_t)ngrid + nbands)))) return rc;  // per band: queue length + queue
    if (do_layout) {
      // (the layout kernel also clears the length of the first band's wide-window queue: one 6 us memset less)
      int rc2 = layout_rows(c, cur, prev, basic, acc_zero, w, h, ch, g.psz, 0, h, 0, h, (uint32_t*)c->wide.p);
      if (rc2) return rc2;
      pl.wide0_zeroed = true;
    }
  }
  if ((rc = upload_tables(c, g.psz))) return rc;
  if ((rc = reserve(c, c->topk, sizeof(uint32_t) * (size_t)ngrid * g.kmax)) ||
      (rc = reserve(c, c->tinfo, sizeof(NlkTarget) * (size_t)ngrid)) ||
      (rc = reserve(c, c->gcoords, sizeof(uint32_t) * (size_t)ngrid * ntagg_alloc)) ||
      (rc = reserve(c, c->marks, sizeof(uint64_t) * (size_t)ngrid)) ||
      (rc = reserve(c, c->active, (size_t)ngrid)))
    return rc;
  mark(c, 1);

  // ---- launch shapes of the block matching
  pl.img_match = basic ? pl.img_basic : pl.img_cur;
  NlkTile& tl = pl.tl;
  // 8 x 4 targets per workgroup, four 4 x 2 blocks that share their squared differences. A small grid
  // (fewer than ~2 such tiles per CU: 256 x 256, 512 x 384, a 34-row strip of a 1080p frame) is latency bound
  // instead: 4 x 2 tiles, one target at a time, two targets per wavefront (C1: match 0.058 -> 0.02 ms).
  // Where the two meet (round 4, temporal / first-frame match ms, blocks against target by target): 384 tiles
  // 0.059 / 0.144 against 0.053 / 0.141; 486: 0.061 / 0.148 against 0.060 / 0.178; 600 (640 x 480): 0.071 / 0.188
  // against 0.071 / 0.221; 950 (800 x 600): 0.087 / 0.241 against 0.095 / 0.302; 1020 (960 x 540, and the 67-row
  // strips of four GPUs): 0.085 / 0.249 against 0.101 / 0.330. (Until then the limit was 1024 tiles.)
  const bool small_grid = (size_t)((g.ngx + 7) / 8) * ((g.ngy + 3) / 4) < 576;
  tl.tgx = nlk_or(c->sw.mtx, small_grid ? 4 : 8);
  tl.tgy = nlk_or(c->sw.mty, small_grid ? 2 : 4);
  // 8 wavefronts per workgroup where the search radius is the temporal one (FLT1 / FLT2 temporal, SMO1), measured at
  // 1080p, match ms with 8 x 8 patches: 8 x 4 tile, 4 wavefronts, blocks of 4 x 2 targets 0.305 (20 wavefronts per CU);
  // 8 x 8 tile, 8 wavefronts, 4 x 2 blocks 0.294 (45 KB: three per CU = 24); 8 x 4 tile, 8 wavefronts with a block of
  // 2 x 2 targets each, 64 registers 0.268 (30 KB: four per CU = 32, a fifth more subtractions and multiplications)
  // - the default. 10 x 10 patches and more: the same 2 x 2 blocks (4K, 12 x 12: 0.864 -> 0.762, 16 x 16: 1.22 -> 0.91).
  // With the spatial radius the tiles are too large for more than the four wavefronts (8 x 8 tile 62 KB: 0.95 -> 1.22).
  // NLK_MATCH_BX2=0 keeps the 4 x 2 blocks, NLK_MATCH_WG8=1 the 8 x 8 tiles.
  tl.threads = NLK_BM_THREADS;
  tl.bx = 4;
  {
    const int halo0 = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
    const bool tiles84 = tl.tgx == 8 && tl.tgy == 4 && (nlk_set(c->sw.match_block) || !small_grid);  // (the tiles of a full-size frame)
    // (the launcher's own condition, match_launch.h: blocks of 2 x 2 exist for MAXM 2, and for MAXM 7 at 8 x 8)
    const int rounds0 = ((2 * halo0 + 1) * (2 * halo0 + 1) + 63) / 64;
    if (nlk_or(c->sw.match_wg8, 0) != 0 && g.psz <= 8 && tiles84) { tl.threads = 512; tl.tgy = 8; }
    else if (nlk_or(c->sw.match_bx2, 1) != 0 && tiles84 && g.psz >= 8 && (rounds0 <= 2 || (g.psz == 8 && rounds0 <= 7))) {
      // (... and the seven-round blocks of a first frame, 441 candidates: 2 x 2 targets per block need 128 registers
      // instead of 168 + spills, two 8-wavefront workgroups per CU instead of three of 4: match 0.947 -> 0.828)
      tl.threads = 512;
      tl.bx = 2;
    }
  }
  tl.ntx = (g.ngx + tl.tgx - 1) / tl.tgx;
  tl.block = nlk_set(c->sw.match_noblock) ? 0 : (nlk_set(c->sw.match_block) ? 1 : !small_grid);
  // the opt-in block-summed distance order (NLK_MATCH_ORDER=block; 8 x 8 patches, k_bm_topk / k_bm_wide); everything
  // else - and the default - sums in the reference's order
  tl.order = (nlk_or(c->sw.match_order, 0) == 1 && g.psz == 8) ? 1 : 0;
  // LDS holds the halo of the dominant window; its row stride = window width
  // (mod 32): candidate i of a window then sits on bank i mod 32, so a
  // wavefront's 64 candidate reads are conflict free
  tl.halo = (g.smoother || g.have_prev) ? g.wsz_t : g.wsz_x;
  {
    const int need = (tl.tgx - 1) * g.step + 2 * tl.halo + g.psz;
    const int wdom = 2 * tl.halo + 1;
    tl.rwp = need + ((wdom - need) % 32 + 32) % 32;
  }
  tl.rh_max = (tl.tgy - 1) * g.step + 2 * tl.halo + g.psz;
  tl.ksel_max = g.kmax;
  auto tile_lds = [&]() {
    return sizeof(float) * ((size_t)ch * tl.rwp * tl.rh_max + 1 + (size_t)(tl.threads / 64) * (3 * tl.ksel_max + ntagg_alloc));
  };
  pl.lds = tile_lds();
  if (pl.lds > 160 * 1024 && tl.threads > NLK_BM_THREADS && tl.tgy == 4) {
    // (very long lists: the per-wavefront list space of 8 wavefronts does not fit beside the tile - 4 wavefronts, 4 x 2 blocks)
    tl.threads = NLK_BM_THREADS;
    tl.bx = 4;
    pl.lds = tile_lds();
  }
  if (pl.lds > 160 * 1024) pl.generic = true;  // (a k or window too large for the tile: the generic kernel)
  // targets of a temporal frame without a valid previous patch search the spatial window
  // (reference: :637); when that one is the wider, they are queued for a second launch
  pl.wide = !pl.generic && g.have_prev && !g.smoother && g.wsz_x > g.wsz_t;
  const int wdom = 2 * tl.halo + 1;
  pl.maxm = (wdom * wdom + 63) / 64;
  if (pl.wide) {
    NlkTile& tw = pl.tw;
    tw = tl;
    const int need = 2 * g.wsz_x + g.psz, ww = 2 * g.wsz_x + 1;
    tw.halo = g.wsz_x;
    tw.rwp = need + ((ww - need) % 32 + 32) % 32;
    tw.rh_max = need;
    const size_t per_wave = (((size_t)ch * tw.rwp * tw.rh_max + 1) & ~(size_t)1) +
                            ((3 * (size_t)tw.ksel_max + ntagg_alloc + 1) & ~(size_t)1);
    pl.lds_w = sizeof(float) * NLK_BM_WAVES * per_wave;
    if (pl.lds_w > 160 * 1024) { pl.generic = true; pl.wide = false; }
    pl.maxm_w = (ncand + 63) / 64;
  }
  c->p_match = pl.img_match; c->p_cur = pl.img_cur; c->p_prev = pl.img_prev;
  c->last = g;
  c->have_last = true;
  return NLK_OK;
}

// phase 1 for the target rows [r0, r0 + rows) of the plan: block matching + selection on `stream`;
// leaves topk / gcoords / tinfo / marks of those rows in the context's buffers
static int match_rows(nlk_ctx* c, const NlkPlan& pl, hipStream_t stream, int r0, int rows, int band) {
  const NlkGeom gb = band_geom(pl.g, r0, rows);
  c->rv = view_rows(c, pl.g, stream, r0, band);
  if (pl.generic) return nlk_launch_match_generic(c, gb, pl.img_match);
  NlkTile tl = pl.tl;
  tl.nty = (rows + tl.tgy - 1) / tl.tgy;
  if (pl.wide && !(pl.wide0_zeroed && r0 == 0 && band == 0))
    HIPCHK(c, hipMemsetAsync(c->rv.wide, 0, sizeof(uint32_t), stream));
  int rc = nlk_launch_match(c, gb, tl, pl.lds, pl.img_match, pl.maxm, false);
  if (rc) return rc;
  if (pl.wide) {
    NlkTile tw = pl.tw;
    tw.nty = tl.nty;
    rc = nlk_launch_match(c, gb, tw, pl.lds_w, pl.img_match, pl.maxm_w, true);
  }
  return rc;
}

// phase 2: replay of the raster-order processed mask over the mark words of the grid rows
// [first, first + rows) (`marks` / `active` = whole-grid arrays). Rows before `first` are context:
// their decisions must be final in `active`.
// `total` = rows of the whole grid when it is replayed band by band (first > 0 continues the band before).
static int commit_rows(nlk_ctx* c, hipStream_t stream, const uint64_t* marks, uint8_t* active, int ngx, int first,
                       int nrows, int R, int total = 0) {
  total = max(total, first + nrows);
  if (R == 0) {
    // a group cannot reach another grid target: nothing is ever skipped
    HIPCHK(c, hipMemsetAsync(active + (size_t)first * ngx, 1, (size_t)nrows * ngx, stream));
    return NLK_OK;
  }
  if (R == 1 && ngx <= 2048 && !nlk_set(c->sw.commit_wave) && !nlk_set(c->sw.commit_lds) && !nlk_set(c->sw.commit_band)) {
    // reach 1: one grid row per step on bit planes (k_commit.h); the arrays cover the whole grid, padded to
    // whole batches (a band's last batch may run into the rows of the next): 4 planes + decisions + row states
    const int rows_pad = (total + NLK_CR_BATCH - 1) / NLK_CR_BATCH * NLK_CR_BATCH + 3 * NLK_CR_BATCH;
    int rc = reserve(c, c->skew, sizeof(uint32_t) * (size_t)rows_pad * 6 * 64);
    if (rc) return rc;
    uint32_t* planes = (uint32_t*)c->skew.p;
    uint32_t* actbits = planes + (size_t)rows_pad * 4 * 64;
    uint32_t* astate = actbits + (size_t)rows_pad * 64;
    hipLaunchKernelGGL(k_marks_planes1, dim3(8, nrows), dim3(256), 0, stream, marks, planes, ngx, first, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_mask_commit_rows1, dim3(1), dim3(64), 0, stream, (const uint32_t*)planes, actbits, astate, ngx,
                       first, nrows);
    hipLaunchKernelGGL(k_active_bytes, dim3((ngx + 255) / 256, nrows), dim3(256), 0, stream, (const uint32_t*)actbits,
                       active, ngx, first);
    HIPCHK(c, hipGetLastError());
    return NLK_OK;
  }
  if (R <= 3 && ngx <= 2048 && !nlk_set(c->sw.commit_wave) && !nlk_set(c->sw.commit_lds) && !nlk_set(c->sw.commit_band)) {
    // reach 2 / 3: the row formulation with the in-row chain solved by iteration (k_commit.h)
    const int np = R + R * (2 * R + 1), pf = 48 / np;
    const int rows_pad = (total + pf - 1) / pf * pf + 3 * pf;
    int rc = reserve(c, c->skew, sizeof(uint32_t) * (size_t)rows_pad * (np + 1 + R) * 64);
    if (rc) return rc;
    uint32_t* planes = (uint32_t*)c->skew.p;
    uint32_t* actbits = planes + (size_t)rows_pad * np * 64;
    uint32_t* astate = actbits + (size_t)rows_pad * 64;
    if (R == 2) {
      hipLaunchKernelGGL(k_marks_planes<2>, dim3(8, nrows), dim3(256), 0, stream, marks, planes, ngx, first, (uint32_t*)nullptr);
      hipLaunchKernelGGL(k_mask_commit_rows<2>, dim3(1), dim3(64), 0, stream, (const uint32_t*)planes, actbits, astate,
                         ngx, first, nrows);
    } else {
      hipLaunchKernelGGL(k_marks_planes<3>, dim3(8, nrows), dim3(256), 0, stream, marks, planes, ngx, first, (uint32_t*)nullptr);
      hipLaunchKernelGGL(k_mask_commit_rows<3>, dim3(1), dim3(64), 0, stream, (const uint32_t*)planes, actbits, astate,
                         ngx, first, nrows);
    }
    hipLaunchKernelGGL(k_active_bytes, dim3((ngx + 255) / 256, nrows), dim3(256), 0, stream, (const uint32_t*)actbits,
                       active, ngx, first);
    HIPCHK(c, hipGetLastError());
    return NLK_OK;
  }
  if (R <= 3 && (first != 0 || !nlk_set(c->sw.commit_lds))) {
    // one lane per grid row, up to 1024 rows per launch; more rows in pieces that start with the
    // previous piece's last R rows as context (k_commit.h)
    int band = nlk_or(c->sw.commit_band, 1024);
    band = band < 2 * R + 2 ? 2 * R + 2 : (band > 1024 ? 1024 : band);
    auto pre = R == 1 ? k_marks_skew<1> : (R == 2 ? k_marks_skew<2> : k_marks_skew<3>);
    auto kern = R == 1 ? k_mask_commit_wave<1> : (R == 2 ? k_mask_commit_wave<2> : k_mask_commit_wave<3>);
    const int end = first + nrows;
    for (int f = first; f < end;) {  // `f` = first row this piece decides
      const int ctx = f == 0 ? 0 : R;
      const int r0 = f - ctx, rows = min(band, end - r0);
      const int threads = ((rows + 63) / 64) * 64;
      const int nsteps = ngx + (R + 1) * (rows - 1);
      const size_t sk_bytes = sizeof(uint32_t) * (size_t)(nsteps + 3 * NLK_CW_PHASE) * threads;
      int rc = reserve(c, c->skew, sk_bytes);
      if (rc) return rc;
      HIPCHK(c, hipMemsetAsync(c->skew.p, 0, sk_bytes, stream));
      hipLaunchKernelGGL(pre, dim3((rows * ngx + 255) / 256), dim3(256), 0, stream, marks + (size_t)r0 * ngx,
                         (uint32_t*)c->skew.p, ngx, rows, threads, (const uint8_t*)active + (size_t)r0 * ngx, ctx);
      hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, stream, (const uint32_t*)c->skew.p,
                         active + (size_t)r0 * ngx, ngx, rows, ctx);
      f = r0 + rows;
    }
    HIPCHK(c, hipGetLastError());
    return NLK_OK;
  }
  if (first != 0) return fail(c, NLK_EINVAL, "the LDS / list mask replays take whole grids only");
  const int ngy = nrows, ngrid = ngx * ngy;
  if (R > 3) {
    // (2R+1)^2 neighbours do not fit the 64-bit mark words: replay from the group-coordinate lists of
    // the match phase that has just run in this context (k_mask_commit_lists)
    if (!c->have_last || c->last.ngx != ngx || c->last.ngy != ngy || c->last.R != R ||
        marks != (const uint64_t*)c->marks.p)
      return fail(c, NLK_EUNSUP, "group reach %d grid cells > 3: the mask replay needs the coordinate lists of the "
                  "context's own match phase (row strips across GPUs are limited to reach 3)", R);
    hipLaunchKernelGGL(k_mask_commit_lists, dim3(1), dim3(1024), 0, stream, (const NlkTarget*)c->tinfo.p,
                       (const uint32_t*)c->gcoords.p, c->last.gstride, active, ngx, ngy, R, c->last.step, c->last.oy);
    HIPCHK(c, hipGetLastError());
    return NLK_OK;
  }
  const int rpt = (ngy + 1023) / 1024;
  const int threads = min(1024, ((ngy + 63) / 64) * 64);
  const size_t bits = sizeof(uint32_t) * ((size_t)(ngrid + (R + 1) * ngx + 63) / 32 + 1);
  if (bits > 160 * 1024 || rpt > 4)
    return fail(c, NLK_EUNSUP, "patch grid %dx%d too large for the mask replay", ngx, ngy);
  auto kern = rpt == 1 ? k_mask_commit<1> : (rpt == 2 ? k_mask_commit<2> : k_mask_commit<4>);
  HIPCHK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bits));
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), bits, stream, marks, active, ngx, ngy, R);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

// The mask replay inside the group kernel's launch (k_group8m.h, NlkGTile::chase): the replay of a grid is a chain
// of ~270 dependent row steps on ONE wavefront (reach 1: 24 us of a 1.1 ms frame at 1080p, reach 2: 80 us of a 2 ms
// first frame, the rest of the chip idle) that stays far ahead of the group kernel's own progress through the rows
// (88 / 300 ns against 2.9 / 3.8 us per grid row) - so workgroup 0 of the group kernel runs it and the others pick
// their targets' decisions up as they are published. Whole-grid, single-band calls of the default kernels only
// (NLK_NO_CHASE=1: the separate kernels; first frames then go back to four bands).
static bool chase_selected(const nlk_ctx* c, const NlkGeom& g) {
  return g.R >= 1 && g.R <= 3 && g.ngx <= 2048 && g.psz == 8 && (g.ch == 1 || g.ch == 3) && g.kmax <= 128 && g.gstride <= 128 &&
         !c->deterministic && c->planes.cap < ((size_t)1 << 32) && !nlk_set(c->sw.no_chase) &&
         !nlk_set(c->sw.commit_wave) && !nlk_set(c->sw.commit_lds) && !nlk_set(c->sw.commit_band) &&
         !nlk_set(c->sw.generic_group) && !nlk_set(c->sw.group_packed) && !nlk_set(c->sw.group_dpp);
}

// bit planes of the grid rows [0, nrows) + a fresh generation of tagged words (the counter lives on the device and
// is advanced by the bit-plane kernel: a step captured into a HIP graph gets a new generation at every replay)
static int chase_prepare(nlk_ctx* c, hipStream_t stream, const uint64_t* marks, int ngx, int nrows, int R) {
  const int np = R == 1 ? 4 : R + R * (2 * R + 1), pf = R == 1 ? NLK_CR_BATCH : 48 / np;
  const int rows_pad = (nrows + pf - 1) / pf * pf + 3 * pf;
  int rc = reserve(c, c->skew, sizeof(uint32_t) * (size_t)rows_pad * np * 64);
  if (rc) return rc;
  const size_t wbytes = sizeof(uint64_t) * (size_t)(nrows + 4 * NLK_CR_BATCH + 1) * 64;
  if (wbytes > c->chase.cap) {
    if ((rc = reserve(c, c->chase, wbytes))) return rc;
    HIPCHK(c, hipMemsetAsync(c->chase.p, 0, c->chase.cap, stream));  // (generation 0; no word of another life carries one)
  }
  uint32_t* planes = (uint32_t*)c->skew.p;
  uint32_t* gen = (uint32_t*)c->chase.p;
  if (R == 1) hipLaunchKernelGGL(k_marks_planes1, dim3(8, nrows), dim3(256), 0, stream, marks, planes, ngx, 0, gen);
  else if (R == 2) hipLaunchKernelGGL(k_marks_planes<2>, dim3(8, nrows), dim3(256), 0, stream, marks, planes, ngx, 0, gen);
  else hipLaunchKernelGGL(k_marks_planes<3>, dim3(8, nrows), dim3(256), 0, stream, marks, planes, ngx, 0, gen);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

// phase 3 for the target rows [r0, r0 + rows) of the last plan (c->last)
static int group_rows(nlk_ctx* c, hipStream_t stream, float* acc, const uint8_t* active_rows, int r0, int rows,
                      int band) {
  const NlkGeom gb = band_geom(c->last, r0, rows);
  c->rv = view_rows(c, c->last, stream, r0, band);
  if (c->chase_on) {
    c->rv.chase_planes = (const uint32_t*)c->skew.p;
    c->rv.chase_words = (uint64_t*)c->chase.p + 64;
    c->rv.chase_gen = (const uint32_t*)c->chase.p;
    c->rv.chase_reach = c->last.R;
    c->rv.chase_row0 = c->chase_row0;
    c->rv.chase_rows = c->chase_rows;
  }
  return launch_group(c, gb, c->p_match, c->p_cur, c->p_prev, acc, active_rows);
}

// Bands of a whole-frame call (NLK_BANDS=<n>, default 1 = off). The mask replay runs on ONE compute
// unit (it is a chain of dependent steps) for ~7 % of a 1080p frame's time. With the grid rows cut in
// bands on two streams, band b+1 is matched while band b's mask is replayed and band b is filtered
// while band b+1's is. Measured at C2 (profiles/README.md): 1.59 ms with one band, 1.64 with two,
// 1.77 with four - the two bands' match kernels share the chip instead of finishing one after the
// other, and the four cross-stream dependencies and the doubled launches cost what the hidden replay
// gains. Kept as an option (and as the exactness test of banding); profiling always runs one band.
static int frame_bands(const nlk_ctx* c, const NlkGeom& g) {
  if (c->profiling || c->deterministic || g.R == 0 || g.R > 3) return 1;  // (deterministic mode: one slab set, one sum order)
  // default: one band for reach 1 (the replay is 2 % of a frame); four for reach 2 / 3, i.e. spatial frames of
  // small patches: their replay was the diagonal one until round 4 (0.28 ms of a first frame at 1080p on one
  // compute unit); as the iterated row replay it takes 0.08 ms, and four bands still win (first frame at 1080p:
  // 2.09-2.10 ms with one band, 2.04-2.10 with two, 2.01-2.05 with four; 2.13 with the diagonal replay in four)
  int nb = nlk_or(c->sw.bands, g.R >= 2 ? 4 : 1);
  nb = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
  while (nb > 1 && g.ngy / nb < 4 * (g.R + 1) + 8) --nb;  // (thin bands: nothing to gain)
  return nb;
}

// `clear`: the accumulator is cleared by the layout kernel (whole-frame calls)
static int frame_accumulate(nlk_ctx* c, float* acc, const float* cur, const float* prev,
                            const float* basic, int w, int h, int ch, float sigma,
                            const struct nlkalman_params* P, int oy, int ngy, int smoother, bool clear) {
  if (!acc) return fail(c, NLK_EINVAL, "null accumulator");
  NlkPlan pl;
  int rc = plan_frame(c, pl, cur, prev, basic, w, h, ch, sigma, P, oy, ngy, smoother, 8, clear ? acc : nullptr);
  if (rc) return rc;
  const NlkGeom& g = pl.g;
  // (a replay that rides inside the group launch needs no bands to hide behind)
  const int nb = (chase_selected(c, g) && !nlk_set(c->sw.bands)) ? 1 : frame_bands(c, g);
  uint8_t* active = (uint8_t*)c->active.p;
  if (nb == 1) {
    if ((rc = match_rows(c, pl, c->stream, 0, g.ngy, 0))) return rc;
    mark(c, 2);
    if (chase_selected(c, g)) {
      if ((rc = chase_prepare(c, c->stream, (const uint64_t*)c->marks.p, g.ngx, g.ngy, g.R))) return rc;
      mark(c, 3);
      c->chase_on = true;  // (group_rows' view of the records carries the planes / words / generation)
      c->chase_row0 = 0; c->chase_rows = g.ngy;
      rc = group_rows(c, c->stream, acc, active, 0, g.ngy, 0);
      c->chase_on = false;
      if (rc) return rc;
      c->lazy_ngx = g.ngx; c->lazy_ngy = g.ngy;  // (the bytes of `active` only when somebody reads the records)
      mark(c, 4);
      return NLK_OK;
    }
    if ((rc = commit_rows(c, c->stream, (const uint64_t*)c->marks.p, active, g.ngx, 0, g.ngy, g.R))) return rc;
    mark(c, 3);
    if ((rc = group_rows(c, c->stream, acc, active, 0, g.ngy, 0))) return rc;
    mark(c, 4);
    return NLK_OK;
  }
  // ---- banded on two streams. Band b runs on stream b & 1; its mask replay waits for band b-1's.
  // (the replay's scratch is sized for the largest piece now: growing it between two bands would
  // synchronise the device)
  {
    const int rows = min(1024, (g.ngy + nb - 1) / nb + g.R + 1), threads = ((rows + 63) / 64) * 64;
    const size_t need = sizeof(uint32_t) * (size_t)(g.ngx + (g.R + 1) * (rows - 1) + 3 * NLK_CW_PHASE) * threads;
    if ((rc = reserve(c, c->skew, need))) return rc;
  }
  hipStream_t st[2] = {c->stream, c->aux_stream};
  hipEvent_t* ev = c->sync_ev;  // [0] layout done, [1] / [2] replay of the last even / odd band done, [3] aux stream done
  HIPCHK(c, hipEventRecord(ev[0], st[0]));
  HIPCHK(c, hipStreamWaitEvent(st[1], ev[0], 0));
  int r0[9];
  for (int b = 0; b <= nb; ++b) r0[b] = (int)((long)g.ngy * b / nb);
  for (int b = 0; b < nb; ++b)
    if ((rc = match_rows(c, pl, st[b & 1], r0[b], r0[b + 1] - r0[b], b))) return rc;
  for (int b = 0; b < nb; ++b) {
    hipStream_t s = st[b & 1];
    if (b > 0) HIPCHK(c, hipStreamWaitEvent(s, ev[1 + ((b - 1) & 1)], 0));
    if ((rc = commit_rows(c, s, (const uint64_t*)c->marks.p, active, g.ngx, r0[b], r0[b + 1] - r0[b], g.R, g.ngy))) return rc;
    HIPCHK(c, hipEventRecord(ev[1 + (b & 1)], s));
    if ((rc = group_rows(c, s, acc, active + (size_t)r0[b] * g.ngx, r0[b], r0[b + 1] - r0[b], b))) return rc;
  }
  HIPCHK(c, hipEventRecord(ev[3], st[1]));
  HIPCHK(c, hipStreamWaitEvent(st[0], ev[3], 0));
  return NLK_OK;
}

int nlk_dev_frame_accumulate(nlk_ctx* c, float* acc, const float* cur, const float* prev,
                             const float* basic, int w, int h, int ch, float sigma,
                             const struct nlkalman_params* P, int oy, int ngy, int smoother) {
  return frame_accumulate(c, acc, cur, prev, basic, w, h, ch, sigma, P, oy, ngy, smoother, false);
}

// ---- the same three phases as separate entry points (exact masks across GPUs)
// rows [r0, r0 + rows) of the strip's target rows (a whole strip: r0 = 0, rows = ngy). _part lays out the pixel
// rows [lay0, lay1) only (planar copies, row test of the validity map) and finishes the validity map of the rows
// [v0, v1): a caller matches the rows that do not depend on a halo still in flight first - laying out its own rows -
// and lays out the halo rows and matches the seam rows once they have arrived (nothing reads rows in flight, nothing
// is laid out twice). _rows lays the whole strip out again on every call.
int nlk_dev_strip_match_part(nlk_ctx* c, const float* cur, const float* prev, const float* basic, int w,
                             int h, int ch, float sigma, const struct nlkalman_params* P, int oy,
                             int ngy, int smoother, int r0, int rows, int lay0, int lay1, int v0, int v1,
                             void* marks_out, int* reach) {
  if (lay0 < 0 || lay1 > h || lay0 > lay1 || v0 < 0 || v1 > h || v0 > v1)
    return fail(c, NLK_EINVAL, "layout rows [%d, %d) / validity rows [%d, %d) outside the %d-row strip", lay0, lay1, v0, v1, h);
  NlkPlan pl;
  int rc = plan_frame(c, pl, cur, prev, basic, w, h, ch, sigma, P, oy, ngy, smoother, 1, nullptr, false);
  if (rc) return rc;
  // (nlk_ctx_set_strip_accumulator: the rows laid out are also the accumulator rows cleared - no separate memset)
  if ((rc = layout_rows(c, cur, prev, basic, c->strip_acc, w, h, ch, pl.g.psz, lay0, lay1, v0, v1))) return rc;
  mark(c, 1);
  if (r0 < 0 || rows < 0 || r0 + rows > pl.g.ngy) return fail(c, NLK_EINVAL, "rows [%d, %d) outside the strip's %d target rows", r0, r0 + rows, pl.g.ngy);
  if (marks_out && pl.g.R > 3)
    return fail(c, NLK_EUNSUP, "group reach %d grid cells > 3: 64-bit mark words cannot describe it (row strips "
                "across GPUs are limited to reach 3; whole-frame calls are not)", pl.g.R);
  // the matcher writes the mark words of these rows straight to their place in the caller's array
  c->marks_ext = (uint64_t*)marks_out;
  rc = rows > 0 ? match_rows(c, pl, c->stream, r0, rows, 0) : NLK_OK;
  c->marks_ext = nullptr;
  if (rc) return rc;
  mark(c, 2);
  if (reach) *reach = c->last.R;
  return NLK_OK;
}

int nlk_dev_strip_match_rows(nlk_ctx* c, const float* cur, const float* prev, const float* basic, int w,
                             int h, int ch, float sigma, const struct nlkalman_params* P, int oy,
                             int ngy, int smoother, int r0, int rows, void* marks_out, int* reach) {
  return nlk_dev_strip_match_part(c, cur, prev, basic, w, h, ch, sigma, P, oy, ngy, smoother, r0, rows, 0, h, 0, h,
                                  marks_out, reach);
}

int nlk_dev_strip_match(nlk_ctx* c, const float* cur, const float* prev, const float* basic, int w,
                        int h, int ch, float sigma, const struct nlkalman_params* P, int oy,
                        int ngy, int smoother, void* marks_out, int* reach) {
  return nlk_dev_strip_match_rows(c, cur, prev, basic, w, h, ch, sigma, P, oy, ngy, smoother, 0, ngy, marks_out, reach);
}

int nlk_ctx_set_strip_accumulator(nlk_ctx* c, float* acc) {
  if (!c) return NLK_EINVAL;
  c->strip_acc = acc;
  return NLK_OK;
}

int nlk_dev_mask_commit(nlk_ctx* c, const void* marks, int ngx, int ngy, int reach,
                        unsigned char* active) {
  if (!c || !marks || !active || ngx < 1 || ngy < 1 || reach < 0)
    return fail(c, NLK_EINVAL, "bad mask-commit arguments");
  NLK_USE_DEVICE(c);
  int rc = commit_rows(c, c->stream, (const uint64_t*)marks, active, ngx, 0, ngy, reach);
  mark(c, 3);
  return rc;
}

int nlk_dev_strip_group(nlk_ctx* c, float* acc, const unsigned char* active) {
  if (!c || !c->have_last) return fail(c, NLK_EINVAL, "nlk_dev_strip_match has not run");
  if (!acc || !active) return fail(c, NLK_EINVAL, "null accumulator / active flags");
  NLK_USE_DEVICE(c);
  int rc = group_rows(c, c->stream, acc, active, 0, c->last.ngy, 0);
  mark(c, 4);
  return rc;
}

// Phases 2 + 3 of a strip in one call: the mask replay over the whole grid's mark words and the strip's groups. Where
// the group kernel can run the replay inside its own launch (chase_selected) only the grid rows down to the strip's
// last one are replayed, by its workgroup 0, and `active` is not written; otherwise exactly nlk_dev_mask_commit +
// nlk_dev_strip_group on `active + gy0 * ngx`.
int nlk_dev_strip_commit_group(nlk_ctx* c, float* acc, const void* marks, int ngx, int ngy, int reach, int gy0,
                               unsigned char* active) {
  if (!c || !c->have_last) return fail(c, NLK_EINVAL, "nlk_dev_strip_match has not run");
  if (!acc || !marks || !active || ngx < 1 || ngy < 1 || reach < 0 || gy0 < 0 || gy0 + c->last.ngy > ngy ||
      ngx != c->last.ngx)
    return fail(c, NLK_EINVAL, "bad strip commit / group arguments");
  NLK_USE_DEVICE(c);
  int rc;
  if (reach == c->last.R && chase_selected(c, c->last)) {
    const int rows = gy0 + c->last.ngy;
    if ((rc = chase_prepare(c, c->stream, (const uint64_t*)marks, ngx, rows, reach))) return rc;
    mark(c, 3);
    c->chase_on = true;
    c->chase_row0 = gy0; c->chase_rows = rows;
    rc = group_rows(c, c->stream, acc, active + (size_t)gy0 * ngx, 0, c->last.ngy, 0);
    c->chase_on = false;
    c->lazy_ngx = ngx; c->lazy_ngy = rows; c->lazy_dst = active;  // (nlk_ctx_flush_active: rows [0, rows) of `active`)
    mark(c, 4);
    return rc;
  }
  if ((rc = commit_rows(c, c->stream, (const uint64_t*)marks, active, ngx, 0, ngy, reach))) return rc;
  mark(c, 3);
  rc = group_rows(c, c->stream, acc, active + (size_t)gy0 * ngx, 0, c->last.ngy, 0);
  mark(c, 4);
  return rc;
}

int nlk_dev_frame_normalize(nlk_ctx* c, float* out, const float* acc, const float* cur, int w,
                            int h, int ch, int y0, int y1) {
  int rc = check_images(c, out, cur, w, h, ch);
  if (rc) return rc;
  if (!acc || y0 < 0 || y1 > h || y0 > y1) return fail(c, NLK_EINVAL, "bad normalise range");
  NLK_USE_DEVICE(c);
  mark(c, 5);
  hipLaunchKernelGGL(ch == 3 ? k_normalize<3> : k_normalize<0>, dim3(2048), dim3(256), 0, c->stream, out, acc, cur, w, h, ch,
                     y0, y1);
  HIPCHK(c, hipGetLastError());
  mark(c, 6);
  return NLK_OK;
}

static int run_frame(nlk_ctx* c, float* out, const float* cur, const float* prev,
                     const float* basic, int w, int h, int ch, float sigma,
                     const struct nlkalman_params* P, int smoother) {
  int rc = check_images(c, out, cur, w, h, ch);
  if (rc) return rc;
  if (!P) return fail(c, NLK_EINVAL, "null parameters");
  if (P->patch_sz < 2) return fail(c, NLK_EUNSUP, "patch size %d not supported", P->patch_sz);
  NLK_USE_DEVICE(c);
  const size_t accb = sizeof(float) * (size_t)w * h * (ch + 1);
  if ((rc = reserve(c, c->acc, accb))) return rc;
  const int step = P->patch_sz / 2;
  if (h < P->patch_sz || w < P->patch_sz) {
    // no target fits (the reference's loops `px < w - psz + 1`, src/nlkalman.c:586-595, do not run): nothing is
    // aggregated and every pixel keeps its input value (:939-942)
    HIPCHK(c, hipMemcpyAsync(out, cur, sizeof(float) * (size_t)w * h * ch, hipMemcpyDeviceToDevice, c->stream));
    return NLK_OK;
  }
  const int ngy = (h - P->patch_sz) / step + 1;
  rc = frame_accumulate(c, (float*)c->acc.p, cur, prev, basic, w, h, ch, sigma, P, 0, ngy, smoother, true);
  if (rc) return rc;
  rc = nlk_dev_frame_normalize(c, out, (const float*)c->acc.p, cur, w, h, ch, 0, h);
  return rc;
}

int nlk_dev_filter_frame(nlk_ctx* c, float* deno1, const float* nisy1, const float* deno0,
                         const float* bsic1, int w, int h, int ch, float sigma,
                         const struct nlkalman_params* P) {
  return run_frame(c, deno1, nisy1, deno0, bsic1, w, h, ch, sigma, P, 0);
}

int nlk_dev_smooth_frame(nlk_ctx* c, float* smoo1, const float* filt1, const float* smoo0,
                         const float* bsic1, int w, int h, int ch, float sigma,
                         const struct nlkalman_params* P) {
  return run_frame(c, smoo1, filt1, smoo0, bsic1, w, h, ch, sigma, P, 1);
}

// ---- the frame functions on HOST images (what the drop-in API of include/nlkalman.h hands over: pageable
// memory, src/nlkalman.h:46-53). A 1080p RGB call moves 75 MB up and 25 MB down over PCIe, as long as its
// kernels run; done one after the other that is 3.2 ms for 1.35 ms of kernels. Here the frame travels in
// row bands: band b is laid out, matched, its mask rows replayed and its groups filtered while band b+1 is
// still on the link (the upload calls return when the host pages are staged, so the host issues them back
// to back and the kernels follow one band behind), and the rows no later band can add to are normalised
// and sent back while the last bands are still being filtered. Results: the banded pipeline's
// (frame_accumulate), i.e. the whole-frame call's decisions and sums.
static int frame_host(nlk_ctx* c, float* out_h, const float* cur_h, const float* prev_h, const float* basic_h, int w,
                      int h, int ch, float sigma, const struct nlkalman_params* P, int smoother) {
  int rc = check_images(c, out_h, cur_h, w, h, ch);
  if (rc) return rc;
  if (!P) return fail(c, NLK_EINVAL, "null parameters");
  NLK_USE_DEVICE(c);
  const size_t npix = (size_t)w * h, bytes = sizeof(float) * npix * ch, rowb = sizeof(float) * (size_t)w * ch;
  if ((rc = reserve(c, c->hw_cur, bytes)) || (rc = reserve(c, c->hw_out, bytes)) ||
      (prev_h && (rc = reserve(c, c->hw_prev, bytes))) || (basic_h && (rc = reserve(c, c->hw_basic, bytes))) ||
      (rc = reserve(c, c->acc, sizeof(float) * npix * (ch + 1))))
    return rc;
  float *d_cur = (float*)c->hw_cur.p, *d_out = (float*)c->hw_out.p;
  float *d_prev = prev_h ? (float*)c->hw_prev.p : nullptr, *d_basic = basic_h ? (float*)c->hw_basic.p : nullptr;
  const int psz = P->patch_sz, step = psz / 2;
  const int ngy = (psz >= 2 && h >= psz) ? (h - psz) / step + 1 : 0;
  const int wall = smoother ? P->search_sz_t : max(P->search_sz_x, P->search_sz_t);  // reach of any window / group
  const int R = ((smoother || prev_h) ? P->search_sz_t : P->search_sz_x) / max(step, 1);
  int nb = nlk_or(c->sw.host_bands, 5);  // (1080p: 2 bands 2.43 ms, 3 2.30, 4 2.25, 5 2.20, 6 2.22; one upload + call + download 2.9)
  nb = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
  while (nb > 1 && ngy / nb < 4 * (R + 1) + 8) --nb;
  if (nb < 2 || R > 3 || c->deterministic || c->profiling || psz > 16 || w < psz || h < psz) {
    // small frames, masks replayed from the coordinate lists, deterministic slabs, per-kernel timing, and
    // everything the frame call rejects: one upload, the whole-frame call, one download
    HIPCHK(c, hipMemcpyAsync(d_cur, cur_h, bytes, hipMemcpyHostToDevice, c->stream));
    if (prev_h) HIPCHK(c, hipMemcpyAsync(d_prev, prev_h, bytes, hipMemcpyHostToDevice, c->stream));
    if (basic_h) HIPCHK(c, hipMemcpyAsync(d_basic, basic_h, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = run_frame(c, d_out, d_cur, d_prev, d_basic, w, h, ch, sigma, P, smoother))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_h, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NLK_OK;
  }
  if (!c->up_stream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&c->dn_stream, hipStreamNonBlocking));
    for (auto& row : c->band_ev)
      for (hipEvent_t& e : row) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  NlkPlan pl;
  float* acc = (float*)c->acc.p;
  if ((rc = plan_frame(c, pl, d_cur, d_prev, d_basic, w, h, ch, sigma, P, 0, ngy, smoother, 8, nullptr, false))) return rc;
  const NlkGeom& g = pl.g;
  uint8_t* active = (uint8_t*)c->active.p;
  // the streams start behind whatever the context's stream was doing (the device images are reused call to call)
  HIPCHK(c, hipEventRecord(c->sync_ev[0], c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->up_stream, c->sync_ev[0], 0));
  HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->sync_ev[0], 0));
  // Bands alternate between two streams, so that the tail of a band's kernels (a quarter of a frame does not
  // fill the chip to the end) overlaps the next band's: E_UP uploaded, E_LAY laid out, E_MASK mask rows
  // replayed, E_GRP groups filtered, E_DONE rows normalised.
  enum { E_UP, E_LAY, E_MASK, E_GRP, E_DONE };
  hipStream_t st[2] = {c->stream, c->aux_stream};
  int up0 = 0, v0 = 0, nz0 = 0, nz[9];
  nz[0] = 0;
  const bool trace = nlk_set(c->sw.host_trace);
  struct timespec ts0, ts1, ts2;
  if (trace) clock_gettime(CLOCK_MONOTONIC, &ts0);
  hipStream_t const home = c->stream;
  for (int b = 0; b < nb; ++b) {
    const int r0 = (int)((long)ngy * b / nb), r1 = (int)((long)ngy * (b + 1) / nb);
    const bool last = b + 1 == nb;
    hipStream_t s = st[b & 1];
    // pixel rows band b reads (windows, patches) and writes (group members): up to up1
    const int up1 = last ? h : min(h, (r1 - 1) * step + wall + psz);
    const size_t off = (size_t)up0 * w * ch;
    HIPCHK(c, hipMemcpyAsync(d_cur + off, cur_h + off, rowb * (up1 - up0), hipMemcpyHostToDevice, c->up_stream));
    if (prev_h) HIPCHK(c, hipMemcpyAsync(d_prev + off, prev_h + off, rowb * (up1 - up0), hipMemcpyHostToDevice, c->up_stream));
    if (basic_h) HIPCHK(c, hipMemcpyAsync(d_basic + off, basic_h + off, rowb * (up1 - up0), hipMemcpyHostToDevice, c->up_stream));
    HIPCHK(c, hipEventRecord(c->band_ev[E_UP][b], c->up_stream));
    HIPCHK(c, hipStreamWaitEvent(s, c->band_ev[E_UP][b], 0));
    if (b > 0) HIPCHK(c, hipStreamWaitEvent(s, c->band_ev[E_LAY][b - 1], 0));  // (the column test reads the rows before)
    const int v1 = last ? h : up1 - psz + 1;
    c->stream = s;  // (the layout helpers launch on the context's stream)
    rc = layout_rows(c, d_cur, d_prev, d_basic, acc, w, h, ch, psz, up0, up1, v0, v1);
    c->stream = home;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->band_ev[E_LAY][b], s));
    if ((rc = match_rows(c, pl, s, r0, r1 - r0, b))) return rc;
    if (b > 0) HIPCHK(c, hipStreamWaitEvent(s, c->band_ev[E_MASK][b - 1], 0));
    if ((rc = commit_rows(c, s, (const uint64_t*)c->marks.p, active, g.ngx, r0, r1 - r0, g.R, g.ngy))) return rc;
    HIPCHK(c, hipEventRecord(c->band_ev[E_MASK][b], s));
    if ((rc = group_rows(c, s, acc, active + (size_t)r0 * g.ngx, r0, r1 - r0, b))) return rc;
    HIPCHK(c, hipEventRecord(c->band_ev[E_GRP][b], s));
    // rows no later band adds to: the groups of the targets from row r1 on start at r1 * step - wall at the earliest
    nz[b + 1] = last ? h : max(nz0, min(h, r1 * step - wall));
    if (nz[b + 1] > nz0) {
      if (b > 0) HIPCHK(c, hipStreamWaitEvent(s, c->band_ev[E_GRP][b - 1], 0));
      hipLaunchKernelGGL(ch == 3 ? k_normalize<3> : k_normalize<0>, dim3(1024), dim3(256), 0, s, d_out, (const float*)acc, (const float*)d_cur, w, h,
                         ch, nz0, nz[b + 1]);
      HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->band_ev[E_DONE][b], s));
    up0 = up1; v0 = v1; nz0 = nz[b + 1];
  }
  if (trace) clock_gettime(CLOCK_MONOTONIC, &ts1);
  // the rows go back band by band: every upload has been issued by now, the kernels of the first bands are done
  for (int b = 0; b < nb; ++b) {
    if (nz[b + 1] <= nz[b]) continue;
    HIPCHK(c, hipStreamWaitEvent(c->dn_stream, c->band_ev[E_DONE][b], 0));
    const size_t off = (size_t)nz[b] * w * ch;
    HIPCHK(c, hipMemcpyAsync(out_h + off, d_out + off, rowb * (nz[b + 1] - nz[b]), hipMemcpyDeviceToHost, c->dn_stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->dn_stream));
  HIPCHK(c, hipStreamSynchronize(c->aux_stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (trace) {
    clock_gettime(CLOCK_MONOTONIC, &ts2);
    fprintf(stderr, "nlk frame_host: %d bands, uploads + launches issued after %.3f ms, downloads done after %.3f ms\n", nb,
            (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6,
            (ts2.tv_sec - ts0.tv_sec) * 1e3 + (ts2.tv_nsec - ts0.tv_nsec) * 1e-6);
  }
  return NLK_OK;
}

int nlk_filter_frame_host(nlk_ctx* c, float* deno1, const float* nisy1, const float* deno0, const float* bsic1, int w,
                          int h, int ch, float sigma, const struct nlkalman_params* P) {
  if (!c) return fail(nullptr, NLK_EINVAL, "null context");
  return frame_host(c, deno1, nisy1, deno0, bsic1, w, h, ch, sigma, P, 0);
}

int nlk_smooth_frame_host(nlk_ctx* c, float* smoo1, const float* filt1, const float* smoo0, const float* bsic1, int w,
                          int h, int ch, float sigma, const struct nlkalman_params* P) {
  if (!c) return fail(nullptr, NLK_EINVAL, "null context");
  return frame_host(c, smoo1, filt1, smoo0, bsic1, w, h, ch, sigma, P, 1);
}

// the tables upload_tables() sends to the device, for tests that pin them (tests/test_fftw_pin.py)
int nlk_host_tables(int psz, float* basis, float* window, float* basis12_regs) {
  if (psz < 2 || psz > 64) return fail(nullptr, NLK_EINVAL, "patch size %d", psz);
  if (basis) host_basis(basis, psz);
  if (window) host_window(window, psz);
  if (basis12_regs)  // the matrix the 12-point flow graph of k_dct12.h (what k_groupp<12> runs) applies: graph(e_j) = column j
    for (int j = 0; j < 12; ++j) {
      float e[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      e[j] = 1.f;
      nlk_host12::nlk_dct12_fast_fwd(e);
      for (int k = 0; k < 12; ++k) basis12_regs[k * 12 + j] = e[k];
    }
  return NLK_OK;
}

// Where a group kernel replayed the processed mask inside its own launch the decisions exist as tagged words only;
// this writes the byte per target of the replayed grid rows into the array the call was given (frame calls: the
// context's own, read by nlk_ctx_read_records) and waits for it. No-op otherwise.
int nlk_ctx_flush_active(nlk_ctx* c) {
  if (!c) return fail(nullptr, NLK_EINVAL, "null context");
  if (!c->lazy_ngx) return NLK_OK;
  NLK_USE_DEVICE(c);
  hipLaunchKernelGGL(k_active_bytes_tagged, dim3((c->lazy_ngx + 255) / 256, c->lazy_ngy), dim3(256), 0, c->stream,
                     (const uint64_t*)c->chase.p + 64, c->lazy_dst ? c->lazy_dst : (uint8_t*)c->active.p, c->lazy_ngx);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->lazy_ngx = c->lazy_ngy = 0;
  c->lazy_dst = nullptr;
  return NLK_OK;
}

int nlk_ctx_read_records(nlk_ctx* c, int* ngrid, int* kmax, int* gmax, unsigned char* active,
                         int* nsel, int* np0, int* nagg, unsigned int* topk,
                         unsigned int* gcoords) {
  if (!c || !c->have_last) return fail(c, NLK_EINVAL, "no frame has been processed");
  NLK_USE_DEVICE(c);
  const NlkGeom& g = c->last;
  const int n = g.ngx * g.ngy;
  if (ngrid) *ngrid = n;
  if (kmax) *kmax = g.kmax;
  if (gmax) *gmax = g.gstride;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (active) {  // (the group kernel replayed the mask itself: expand its decision words now)
    int rc = nlk_ctx_flush_active(c);
    if (rc) return rc;
  }
  if (active) HIPCHK(c, hipMemcpy(active, c->active.p, n, hipMemcpyDeviceToHost));
  if (nsel || np0 || nagg) {
    NlkTarget* t = (NlkTarget*)malloc(sizeof(NlkTarget) * n);
    if (!t) return fail(c, NLK_ENOMEM, "host malloc failed");
    hipError_t e = hipMemcpy(t, c->tinfo.p, sizeof(NlkTarget) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      for (int i = 0; i < n; ++i) {
        if (nsel) nsel[i] = t[i].nsel;
        if (np0) np0[i] = t[i].np0;
        if (nagg) nagg[i] = t[i].nagg;
      }
    free(t);
    HIPCHK(c, e);
  }
  if (topk)
    HIPCHK(c, hipMemcpy(topk, c->topk.p, sizeof(uint32_t) * (size_t)n * g.kmax,
                        hipMemcpyDeviceToHost));
  if (gcoords)
    HIPCHK(c, hipMemcpy(gcoords, c->gcoords.p, sizeof(uint32_t) * (size_t)n * g.gstride,
                        hipMemcpyDeviceToHost));
  return NLK_OK;
}

}  // extern "C"

