// tvl1_host.h — host side of the TV-L1 optical flow entry points of include/nlk_hip.h
// (included by nlk_hip.hip; kernels: k_tvl1.h). Reference: lib/tvl1flow/tvl1flow_lib.c:345-474
// (multiscale driver), :93-275 (one scale), main.c:26-35, 152-157 (defaults, scale count).
#pragma once

namespace {

// normalised half kernel, exactly as the reference computes it (mask.c:229-256)
int tv_gauss_kernel(nlk_ctx* c, double sigma, int nx, int ny, NlkTvGauss* g) {
  const int rad = (int)(5 * sigma) + 1;
  if (rad > 32) return fail(c, NLK_EUNSUP, "TV-L1: Gaussian sigma %.3f too large (radius %d > 32)", sigma, rad);
  if (rad > nx || rad > ny)
    return fail(c, NLK_EUNSUP, "TV-L1: a %dx%d pyramid level is smaller than the Gaussian radius %d", nx, ny, rad);
  const double den = 2 * sigma * sigma;
  for (int i = 0; i < rad; ++i) g->b[i] = 1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den);
  double norm = 0;
  for (int i = 0; i < rad; ++i) norm += g->b[i];
  norm *= 2;
  norm -= g->b[0];
  for (int i = 0; i < rad; ++i) g->b[i] /= norm;
  g->rad = rad;
  return NLK_OK;
}

inline dim3 tv_grid(int nx, int ny, int images = 1) { return dim3((nx + 31) / 32, (ny + 7) / 8, images); }
const dim3 tv_block(32, 8);

// two images at once: in -> out (may alias), tmp = scratch of the same size
int tv_gaussian2(nlk_ctx* c, const float* in_a, float* out_a, float* tmp_a, const float* in_b, float* out_b,
                 float* tmp_b, int nx, int ny, double sigma) {
  NlkTvGauss g;
  int rc = tv_gauss_kernel(c, sigma, nx, ny, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tv_gauss, tv_grid(nx, ny, 2), tv_block, 0, c->stream, in_a, tmp_a, in_b, tmp_b, nx, ny, g, 0);
  hipLaunchKernelGGL(k_tv_gauss, tv_grid(nx, ny, 2), tv_block, 0, c->stream, (const float*)tmp_a, out_a,
                     (const float*)tmp_b, out_b, nx, ny, g, 1);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

// waits until the launch that closes a group has posted the solver state with sequence number `seq`
// (NlkTvMail): a spin on host memory, with a look at the stream now and then so that a failed
// launch cannot hang the caller
int tv_wait_mail(nlk_ctx* c, unsigned seq) {
  // system-scope acquire load pairing with the kernel's system-scope release store of `seq`
  // (k_tvl1.h: nlk_tv_post): the solver state posted before it is visible once the number matches
  unsigned* flag = &c->tv_host->seq;
  auto posted = [&]() { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq; };
  for (unsigned spins = 0;; ++spins) {
    if (posted()) break;
    if ((spins & 0xFFF) == 0xFFF) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {  // everything has run
        if (posted()) break;
        return fail(c, NLK_EHIP, "TV-L1: the solver state was not posted");
      }
      if (q != hipErrorNotReady) return fail(c, NLK_EHIP, "TV-L1: %s", hipGetErrorString(q));
    }
    __builtin_ia32_pause();
  }
  return NLK_OK;
}

// one scale (reference: tvl1flow_lib.c:93-275)
int tv_scale(nlk_ctx* c, const float* I0, const float* I1, float* u1, float* u2, int nx, int ny,
             const nlk_tvl1_params& P, float* work, float* alt, float* part, NlkTvState* st) {
  const size_t n = (size_t)nx * ny;
  NlkTvLevel L;
  L.I0 = I0; L.I1 = I1; L.u1 = u1; L.u2 = u2;
  L.I1x = work; L.I1y = L.I1x + n; L.I1wx = L.I1y + n; L.I1wy = L.I1wx + n; L.grad = L.I1wy + n;
  L.rho_c = L.grad + n; L.p11 = L.rho_c + n; L.p12 = L.p11 + n; L.p21 = L.p12 + n; L.p22 = L.p21 + n;
  L.part = part; L.st = st; L.mail = c->tv_host;
  L.nx = nx; L.ny = ny; L.nwarps = P.nwarps;
  L.l_t = P.lambda * P.theta; L.theta = P.theta; L.taut = P.tau / P.theta; L.eps2 = P.epsilon * P.epsilon;
  size_t wg_max = nlk_set(c->sw.tv_wg_pixels) ? (size_t)c->sw.tv_wg_pixels : NLK_TV_WG_PIXELS;
  if (wg_max > NLK_TV_WG_PIXELS) wg_max = NLK_TV_WG_PIXELS;  // (the kernel's LDS arrays)
  if (n <= wg_max) {  // the whole level inside one workgroup
    // (round 6) no more wavefronts than the level has pixels for: thread t owns pixel t (+ 1024 m), so a level of up to
    // 1024 pixels gets the same pixel -> lane mapping from ceil(n / 64) wavefronts as from 16, the workgroup sum adds
    // the same wavefront sums in the same order (the idle wavefronts only ever added zeros: bit-identical), and every
    // barrier and the serial sum over the wavefronts cost a fraction - the 30 x 17 and 15 x 8 levels of a 1080p flow
    // are chains of such latencies (~1.9 us an iteration with 16 wavefronts).
    const int wg_threads = n <= (size_t)NLK_TV_THREADS ? (int)((n + 63) / 64) * 64 : NLK_TV_THREADS;
    hipLaunchKernelGGL(k_tv_level_wg, dim3(1), dim3(nlk_set(c->sw.tv_wg_full) ? NLK_TV_THREADS : wg_threads), 0, c->stream, L);
    HIPCHK(c, hipGetLastError());
    return NLK_OK;
  }
  const dim3 grid((nx + 63) / 64, (ny + 3) / 4);
  const int nparts = grid.x * grid.y;
  hipLaunchKernelGGL(k_tv_init, grid, dim3(256), 0, c->stream, L);
  if (nlk_set(c->sw.tv_unblocked)) {  // one launch per half iteration (kept for comparison)
    const int batch = nlk_or(c->sw.tv_batch, 12);
    for (int wi = 0; wi < P.nwarps; ++wi) {
      hipLaunchKernelGGL(k_tv_warp, grid, dim3(256), 0, c->stream, L);
      int launched = 0;
      while (launched < NLK_TV_MAXIT) {
        const int upto = launched + batch < NLK_TV_MAXIT ? launched + batch : NLK_TV_MAXIT;
        for (int it = launched + 1; it <= upto; ++it) {
          hipLaunchKernelGGL(k_tv_primal, grid, dim3(256), 0, c->stream, L, it);
          hipLaunchKernelGGL(k_tv_dual, grid, dim3(256), 0, c->stream, L, it, nparts);
        }
        launched = upto;
        HIPCHK(c, hipMemcpyAsync(&c->tv_host->st, st, sizeof(NlkTvState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->tv_host->st.stop_iter < NLK_TV_MAXIT || c->tv_host->st.last >= NLK_TV_MAXIT) break;
      }
    }
    HIPCHK(c, hipGetLastError());
    return NLK_OK;
  }
  // blocked driver: NLK_TV_K iterations per launch between two state buffers. A = the level's
  // own arrays, B = a second set; the host launches batches ahead and reads the state every
  // few batches (each batch judges its predecessor's convergence when it starts, the launch that
  // closes a group judges the last one); batches after the converged one are no-ops, so the final state sits in the
  // buffer written by batch ceil(stop / K).
  NlkTvBuf A = {u1, u2, L.p11, L.p12, L.p21, L.p22};
  NlkTvBuf B = {alt, alt + n, alt + 2 * n, alt + 3 * n, alt + 4 * n, alt + 5 * n};
  // tile shape by the number of tiles: tall tiles do a quarter less halo work but need enough tiles to
  // keep every CU busy; with at least two tall tiles per CU, workgroups of 512 (two per CU) let one
  // tile load or store while the other computes
  const int nb16 = ((nx + 63) / 64) * ((ny + NLK_TV_TH - 1) / NLK_TV_TH), nb32 = ((nx + 63) / 64) * ((ny + NLK_TV_TH2 - 1) / NLK_TV_TH2);
  int shape = nb32 >= 512 ? 2 : nb16 >= 400 ? nlk_or(c->sw.tv_mid, 1) : 0;
  // levels that leave most of the chip idle (fewer than 100 tiles of 64 x 16): tiles of 16 x 16 in workgroups of 576
  // (the 24 x 24 region, a pixel per thread) - a launch there is a chain of latencies (judge, load, 8 half iterations
  // between barriers, store), shorter with 9 wavefronts per workgroup than with 16; 1080p flow at fscale 1: 3.20 ->
  // 3.02 ms (profiles/README.md round 5: six shapes per level class; levels of 100-400 tiles do not respond)
  if (shape == 0 && nb16 < 100) shape = 4;
  shape = nlk_or(c->sw.tv_shape, shape);
  const bool deep = shape == 0 && nlk_set(c->sw.tv_deep);  // (8 iterations per launch, shape 0 only: measured slower, kept for experiments)
  const int K = deep ? NLK_TV_K2 : NLK_TV_K;
  // shape -> tile width, tile height, threads, kernel
  struct Shape { int tw, th, bt; decltype(&k_tv_block<64, NLK_TV_TH, NLK_TV_BT, NLK_TV_K>) kern; };
  static const Shape shapes[] = {
    {64, NLK_TV_TH, NLK_TV_BT, k_tv_block<64, NLK_TV_TH, NLK_TV_BT, NLK_TV_K>},      // 0
    {64, NLK_TV_TH2, NLK_TV_BT, k_tv_block<64, NLK_TV_TH2, NLK_TV_BT, NLK_TV_K>},    // 1
    {64, NLK_TV_TH2, NLK_TV_BT2, k_tv_block<64, NLK_TV_TH2, NLK_TV_BT2, NLK_TV_K>},  // 2
    {64, NLK_TV_TH, NLK_TV_BT2, k_tv_block<64, NLK_TV_TH, NLK_TV_BT2, NLK_TV_K>},    // 3
    {16, 16, 576, k_tv_block<16, 16, 576, NLK_TV_K>},                                // 4
  };
  if (shape < 0 || shape >= (int)(sizeof(shapes) / sizeof(shapes[0]))) shape = 0;
  const Shape& sh = shapes[shape];
  const int th = sh.th, bt = sh.bt;
  const auto block_kernel = deep ? k_tv_block<64, NLK_TV_TH, NLK_TV_BT, NLK_TV_K2> : sh.kern;
  const auto decide_kernel = deep ? k_tv_decide<NLK_TV_K2> : k_tv_decide<NLK_TV_K>;
  const dim3 bgrid((nx + sh.tw - 1) / sh.tw, (ny + th - 1) / th);
  const int nblocks = bgrid.x * bgrid.y;
  const bool inline_judge = nblocks <= nlk_or(c->sw.tv_inline, 600);
  // batches between two looks at the state (a warp needs 5 to 50 iterations, mostly under 16)
  const int look = deep ? nlk_or(c->sw.tv_look2, 2)
                        : nlk_or(c->sw.tv_look, 4);
  for (int wi = 0; wi < P.nwarps; ++wi) {
    hipLaunchKernelGGL(k_tv_warp, grid, dim3(256), 0, c->stream, L);
    NlkTvBuf cur = {L.u1, L.u2, L.p11, L.p12, L.p21, L.p22};
    const bool cur_is_a = cur.u1 == A.u1;
    NlkTvBuf oth = cur_is_a ? B : A;
    const NlkTvBuf first = cur, second = oth;  // buffers of the warp's first batch
    int n0 = 0, batches = 0, last_n0 = 0, last_count = 0;
    while (n0 < NLK_TV_MAXIT) {
      for (int q = 0; q < look && n0 < NLK_TV_MAXIT; ++q) {
        const int count = NLK_TV_MAXIT - n0 < K ? NLK_TV_MAXIT - n0 : K;
        hipLaunchKernelGGL(block_kernel, bgrid, dim3(bt), 0, c->stream, L, cur, oth, n0, count,
                           inline_judge && q ? 2 : 0, 0u);
        if (!inline_judge) hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(256), 0, c->stream, L, n0, count, nblocks);
        last_n0 = n0;
        last_count = count;
        const NlkTvBuf t = cur; cur = oth; oth = t;
        n0 += count;
        ++batches;
      }
      // the batch that ran past the stop (if any) is redone from its input, once per group
      // (it also judges the group's last batch)
      const unsigned seq = ++c->tv_seq;
      hipLaunchKernelGGL(block_kernel, bgrid, dim3(bt), 0, c->stream, L, first, second, last_n0, last_count,
                         inline_judge ? 1 : 3, seq);
      int rc = tv_wait_mail(c, seq);
      if (rc) return rc;
      if (c->tv_host->st.stop_iter < NLK_TV_MAXIT || c->tv_host->st.fin_stop < NLK_TV_MAXIT ||
          c->tv_host->st.last >= NLK_TV_MAXIT)
        break;
    }
    // where the state of iteration `last` lives: written by batch ceil(last / K), batches alternate
    const int used = (c->tv_host->st.last + K - 1) / K;
    const bool final_is_start = (used % 2) == 0;
    const NlkTvBuf fin = final_is_start ? (cur_is_a ? A : B) : (cur_is_a ? B : A);
    L.u1 = fin.u1; L.u2 = fin.u2; L.p11 = fin.p11; L.p12 = fin.p12; L.p21 = fin.p21; L.p22 = fin.p22;
    if (nlk_set(c->sw.tv_trace))
      fprintf(stderr, "tvl1 %dx%d warp %d: %d iterations, %d batches launched\n", nx, ny, wi, c->tv_host->st.last, batches);
  }
  if (L.u1 != u1) {  // the level's flow belongs in the pyramid arrays
    HIPCHK(c, hipMemcpyAsync(u1, L.u1, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(u2, L.u2, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

}  // namespace

extern "C" {

void nlk_tvl1_default_params(struct nlk_tvl1_params* p) {  // reference: lib/tvl1flow/main.c:26-35
  p->tau = 0.25f; p->lambda = 0.15f; p->theta = 0.3f;
  p->nscales = 100; p->fscale = 0; p->zfactor = 0.5f;
  p->nwarps = 5; p->epsilon = 0.01f;
}

int nlk_tvl1_scales(int w, int h, int nscales, float zfactor) {  // reference: main.c:152-157
  const float N = 1 + log(hypot(w, h) / 16.0) / log(1 / zfactor);
  return N < nscales ? (int)N : nscales;
}

int nlk_dev_gray(nlk_ctx* c, float* gray, const float* im, int w, int h, int ch) {
  if (!c || !gray || !im || w <= 0 || h <= 0 || ch <= 0) return fail(c, NLK_EINVAL, "nlk_dev_gray: bad argument");
  NLK_USE_DEVICE(c);
  const int n = w * h;
  hipLaunchKernelGGL(k_tv_gray, dim3((n + 255) / 256), dim3(256), 0, c->stream, im, gray, n, ch);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

int nlk_dev_occlusion_mask(nlk_ctx* c, float* mask, const float* flow, int w, int h, float th) {
  if (!c || !mask || !flow || w <= 0 || h <= 0) return fail(c, NLK_EINVAL, "nlk_dev_occlusion_mask: bad argument");
  NLK_USE_DEVICE(c);
  hipLaunchKernelGGL(k_tv_occlusion, tv_grid(w, h), tv_block, 0, c->stream, flow, mask, w, h, th);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

int nlk_dev_tvl1_flow(nlk_ctx* c, float* flow, const float* I0, const float* I1, int w, int h,
                      const struct nlk_tvl1_params* P, int* iterations) {
  if (!c || !flow || !I0 || !I1 || !P) return fail(c, NLK_EINVAL, "nlk_dev_tvl1_flow: null argument");
  if (w < 2 || h < 2) return fail(c, NLK_EINVAL, "nlk_dev_tvl1_flow: %dx%d image", w, h);
  if (!(P->tau > 0) || !(P->lambda > 0) || !(P->theta > 0) || !(P->zfactor > 0 && P->zfactor < 1) ||
      P->nscales < 1 || P->nscales > 64 || P->fscale < 0 || P->nwarps < 1 || !(P->epsilon > 0))
    return fail(c, NLK_EINVAL, "nlk_dev_tvl1_flow: parameter out of range");
  NLK_USE_DEVICE(c);
  const int ns = P->nscales;
  int W[64], H[64];
  W[0] = w; H[0] = h;
  size_t pyr = (size_t)w * h;
  for (int s = 1; s < ns; ++s) {  // reference: zoom.c:23-35
    W[s] = (int)((float)W[s - 1] * P->zfactor + 0.5);
    H[s] = (int)((float)H[s - 1] * P->zfactor + 0.5);
    if (W[s] < 2 || H[s] < 2) return fail(c, NLK_EINVAL, "nlk_dev_tvl1_flow: %d scales leave a %dx%d level", ns, W[s], H[s]);
    pyr += (size_t)W[s] * H[s];
  }
  const size_t n0 = (size_t)w * h;
  // scratch: 4 pyramids (I0, I1, u1, u2) + 10 work images + 2 temporaries at full size + partial sums
  // workgroups of an iteration kernel at full size: 64 x 4 pixels each (k_tv_primal), or the smallest tiles of the
  // blocked kernel (16 x 16: tv_scale's table)
  const size_t nparts_a = (size_t)((w + 63) / 64) * ((h + 3) / 4), nparts_b = (size_t)((w + 15) / 16) * ((h + 15) / 16);
  const size_t nparts0 = nparts_a > nparts_b ? nparts_a : nparts_b;
  const size_t floats = 4 * pyr + 18 * n0 + 2 * NLK_TV_K2 * nparts0 + 64;
  int rc = reserve(c, c->tv, sizeof(float) * floats);
  if (rc) return rc;
  if (!c->tv_host) {
    HIPCHK(c, hipHostMalloc((void**)&c->tv_host, sizeof(NlkTvMail)));
    memset(c->tv_host, 0, sizeof(NlkTvMail));
  }
  float* base = (float*)c->tv.p;
  float *I0s[64], *I1s[64], *U1[64], *U2[64];
  float* q = base;
  for (int s = 0; s < ns; ++s) {
    const size_t n = (size_t)W[s] * H[s];
    I0s[s] = q; q += n; I1s[s] = q; q += n; U1[s] = q; q += n; U2[s] = q; q += n;
  }
  float* work = q; q += 10 * n0;
  float* tmp = q; q += n0;
  float* tmp2 = q; q += n0;
  float* alt = q; q += 6 * n0;  // second state buffer of the blocked driver
  float* part = q; q += 2 * NLK_TV_K2 * nparts0;  // two batches' partial sums
  NlkTvState* st = (NlkTvState*)q;  // 64 floats reserved
  int* mm = (int*)(q + 8);

  // normalise both images to 0..255 with one common range, pre-smooth (reference: :376-381)
  HIPCHK(c, hipMemsetAsync(st, 0, sizeof(NlkTvState), c->stream));
  hipLaunchKernelGGL(k_tv_init_minmax, dim3(1), dim3(1), 0, c->stream, mm);
  hipLaunchKernelGGL(k_tv_minmax, dim3(512), dim3(1024), 0, c->stream, I0, I1, (int)n0, mm);
  hipLaunchKernelGGL(k_tv_normalize, dim3((n0 + 255) / 256), dim3(256), 0, c->stream, I0, I1, I0s[0],
                     I1s[0], (int)n0, (const int*)mm);
  float *tmp_b = work, *tmp2_b = work + n0;  // (the level arrays are idle while the pyramid is built)
  if ((rc = tv_gaussian2(c, I0s[0], I0s[0], tmp, I1s[0], I1s[0], tmp_b, w, h, 0.8))) return rc;
  // pyramid (reference: :384-398, zoom.c:44-79)
  const float zsigma = 0.6 * sqrt(1.0 / (P->zfactor * P->zfactor) - 1.0);
  for (int s = 1; s < ns; ++s) {
    if ((rc = tv_gaussian2(c, I0s[s - 1], tmp2, tmp, I1s[s - 1], tmp2_b, tmp_b, W[s - 1], H[s - 1], zsigma))) return rc;
    hipLaunchKernelGGL(k_tv_zoom, tv_grid(W[s], H[s], 2), tv_block, 0, c->stream, (const float*)tmp2, I0s[s],
                       (const float*)tmp2_b, I1s[s], W[s - 1], H[s - 1], W[s], H[s], P->zfactor, P->zfactor, 1.f, 0);
  }
  const size_t nc = (size_t)W[ns - 1] * H[ns - 1];
  HIPCHK(c, hipMemsetAsync(U1[ns - 1], 0, sizeof(float) * nc, c->stream));
  HIPCHK(c, hipMemsetAsync(U2[ns - 1], 0, sizeof(float) * nc, c->stream));
  // coarse to fine; scales finer than fscale only receive the upsampled flow (reference: :404-461)
  const float inv = (float)1.0 / P->zfactor;
  for (int s = ns - 1; s >= 0; --s) {
    if (s >= P->fscale)
      if ((rc = tv_scale(c, I0s[s], I1s[s], U1[s], U2[s], W[s], H[s], *P, work, alt, part, st))) return rc;
    if (s == 0) break;
    const float fx = (float)W[s - 1] / W[s], fy = (float)H[s - 1] / H[s];  // zoom.c:94-95
    hipLaunchKernelGGL(k_tv_zoom, tv_grid(W[s - 1], H[s - 1], 2), tv_block, 0, c->stream, (const float*)U1[s],
                       U1[s - 1], (const float*)U2[s], U2[s - 1], W[s], H[s], W[s - 1], H[s - 1], fx, fy, inv, 1);
  }
  hipLaunchKernelGGL(k_tv_interleave, dim3((n0 + 255) / 256), dim3(256), 0, c->stream, (const float*)U1[0],
                     (const float*)U2[0], flow, (int)n0);
  HIPCHK(c, hipGetLastError());
  if (iterations) {  // (the only read-back, and only when asked for)
    HIPCHK(c, hipMemcpyAsync(&c->tv_host->st, st, sizeof(NlkTvState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *iterations = c->tv_host->st.iters;
  }
  return NLK_OK;
}

}  // extern "C"
