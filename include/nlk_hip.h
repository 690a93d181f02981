/* nlk_hip.h — thin C-ABI over the hand-written HIP (gfx950) kernels.
 *
 * This is the boundary a maintainer of the reference would bind to replace the
 * body of its three frame functions and its three image helpers. Plain
 * pointers and sizes only; every entry point returns 0 on success or a
 * negative NLK_E* code, with a message retrievable by nlk_last_error().
 * Device pointers are ordinary HIP device addresses (hipMalloc / a torch
 * tensor's data_ptr()); images keep the reference's HWC interleaved float32
 * layout, index (x + y*w)*ch + c (reference: src/nlkalman.c:555-560).
 *
 * What each entry point replaces in the reference:
 *   nlk_dev_rgb2opp / nlk_dev_opp2rgb   src/nlkalman.c:92-130
 *   nlk_dev_warp_bicubic                src/nlkalman.c:29-88
 *   nlk_dev_filter_frame                src/nlkalman.c:518-951  (nlkalman_filter_frame)
 *   nlk_dev_smooth_frame                src/nlkalman.c:1409-1865 (nlkalman_smooth_frame)
 *   nlk_dev_strip_match / nlk_dev_mask_commit / nlk_dev_strip_group
 *                                       the three phases of the frame functions:
 *                                       :605-857 (search, selection, groups), :597-600 +
 *                                       :930-931 (processed mask), :713-932 (filtering)
 *   nlk_dev_frame_accumulate/_normalize the same two functions split at
 *                                       src/nlkalman.c:939 / :1853 so that row
 *                                       strips can exchange accumulator halos
 *   nlk_dev_tvl1_flow                   lib/tvl1flow/tvl1flow_lib.c:345-474
 *                                       (Dual_TVL1_optic_flow_multiscale)
 *   nlk_tvl1_default_params / _scales   lib/tvl1flow/main.c:26-35, 152-157
 *   nlk_dev_gray                        lib/iio/iio.c:1048-1056 (what the flow tool's
 *                                       reader does to a colour image)
 *   nlk_dev_occlusion_mask              scripts/nlkalman-seq.sh:70-73 (plambda)
 *   nlk_dev_image_dct / nlk_dev_copy_block  lib/multiscale/multiscaler.cpp:21-107 and the
 *                                       coefficient copies of decompose / recompose
 */
#ifndef NLK_HIP_H
#define NLK_HIP_H

#include <stddef.h>
#include "nlkalman.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  NLK_OK = 0,
  NLK_ENODEV = -1,   /* no HIP device / device index out of range */
  NLK_EHIP = -2,     /* a HIP runtime call failed */
  NLK_EINVAL = -3,   /* bad argument (NULL image, non-positive size, ...) */
  NLK_EUNSUP = -4,   /* parameter combination the kernels do not cover */
  NLK_ENOMEM = -5
};

typedef struct nlk_ctx nlk_ctx; /* one per (process, device): stream + scratch */

/* per-kernel device time of a frame call, milliseconds (HIP events recorded on
 * the context's stream around each kernel; only when profiling is enabled) */
struct nlk_timings {
  float layout_ms;    /* HWC -> planar copies + validity map */
  float match_ms;     /* block matching + k-NN selection */
  float commit_ms;    /* processed-mask replay */
  float group_ms;     /* DCT + statistics + shrinkage + IDCT + aggregation */
  float normalize_ms; /* accumulator normalisation */
  float total_ms;
};

int nlk_device_count(void);
int nlk_ctx_create(nlk_ctx **ctx, int device);
void nlk_ctx_destroy(nlk_ctx *ctx);
const char *nlk_last_error(const nlk_ctx *ctx); /* ctx may be NULL: last global error */
int nlk_ctx_set_profiling(nlk_ctx *ctx, int on);
/* mean over the frame calls made since nlk_ctx_set_profiling(ctx, 1); synchronises */
int nlk_ctx_get_timings(nlk_ctx *ctx, struct nlk_timings *t);
/* Deterministic aggregation: two calls on the same inputs give bit-identical outputs. By default the
 * group kernels add their accumulator tiles to the frame with global float atomics, whose order
 * varies from run to run (as the reference's `omp atomic` adds do, src/nlkalman.c:923-931); with
 * this switch every workgroup writes its tile to a slab of its own and a gather kernel sums the
 * slabs in a fixed order. Also set by NLK_DETERMINISTIC=1 in the environment at context creation. */
int nlk_ctx_set_deterministic(nlk_ctx *ctx, int on);
/* The NLK_* environment switches (variants for comparison tests and experiments: DESIGN.md appendix) are read
 * once, by nlk_ctx_create. This reads them again - for `ctx`, or for every live context of the process when
 * ctx is NULL (the test suite changes the environment under live contexts). */
int nlk_ctx_reload_switches(nlk_ctx *ctx);
/* run the context's work on an externally owned hipStream_t; NULL is the legacy
 * default stream itself (what torch.cuda.current_stream() is unless changed), so
 * that the kernels order with the caller's own work on that stream.
 * nlk_ctx_use_own_stream switches back to the context's private stream. */
int nlk_ctx_set_stream(nlk_ctx *ctx, void *hip_stream);
int nlk_ctx_use_own_stream(nlk_ctx *ctx);
void *nlk_ctx_get_stream(nlk_ctx *ctx);

/* device memory + transfers (so that C callers need no HIP headers) */
int nlk_dev_alloc(nlk_ctx *ctx, void **dptr, size_t bytes);
int nlk_dev_free(nlk_ctx *ctx, void *dptr);
int nlk_h2d(nlk_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int nlk_d2h(nlk_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int nlk_d2d(nlk_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
int nlk_sync(nlk_ctx *ctx);
/* building blocks of the row-strip split across devices (host/multidev.c): clear, dst += src on
 * device memory, and a copy between two contexts' devices (hipMemcpyPeerAsync on dst_ctx's stream,
 * after everything enqueued so far on src_ctx's stream) */
int nlk_dev_zero(nlk_ctx *ctx, void *dptr, size_t bytes);
int nlk_dev_add(nlk_ctx *ctx, float *dst, const float *src, size_t count);
int nlk_dev_copy_peer(nlk_ctx *dst_ctx, void *dst, nlk_ctx *src_ctx, const void *src, size_t bytes);
/* page-locked host memory: transfers from / to it run at the link's rate instead of through a
 * staging copy (file-based callers that keep a pool of frame buffers: host/main_seq.c) */
int nlk_host_alloc(nlk_ctx *ctx, void **hptr, size_t bytes);
int nlk_host_free(nlk_ctx *ctx, void *hptr);

/* image helpers on device-resident HWC images (asynchronous on the ctx stream) */
int nlk_dev_rgb2opp(nlk_ctx *ctx, float *im, int w, int h, int ch);
int nlk_dev_opp2rgb(nlk_ctx *ctx, float *im, int w, int h, int ch);
int nlk_dev_warp_bicubic(nlk_ctx *ctx, float *imw, const float *im,
                         const float *of, const float *msk, int w, int h, int ch);

/* whole-frame hot path on device-resident images; deno0/bsic1 may be NULL */
int nlk_dev_filter_frame(nlk_ctx *ctx, float *deno1, const float *nisy1,
                         const float *deno0, const float *bsic1, int w, int h,
                         int ch, float sigma, const struct nlkalman_params *prms);
int nlk_dev_smooth_frame(nlk_ctx *ctx, float *smoo1, const float *filt1,
                         const float *smoo0, const float *bsic1, int w, int h,
                         int ch, float sigma, const struct nlkalman_params *prms);

/* the same two functions on HOST images (pageable memory, what src/nlkalman.h:46-53 hands over): the frame
 * travels over PCIe in row bands while the bands before it are matched and filtered, and the finished rows
 * travel back while the last bands are filtered. Synchronous: the output is complete on return. */
int nlk_filter_frame_host(nlk_ctx *ctx, float *deno1, const float *nisy1, const float *deno0,
                          const float *bsic1, int w, int h, int ch, float sigma,
                          const struct nlkalman_params *prms);
int nlk_smooth_frame_host(nlk_ctx *ctx, float *smoo1, const float *filt1, const float *smoo0,
                          const float *bsic1, int w, int h, int ch, float sigma,
                          const struct nlkalman_params *prms);

/* ---- optical flow between two frames: the dual TV-L1 method the
 * pipelines run before every filter call (scripts/nlkalman-seq.sh:57-66). Images are
 * single-channel float (w*h); `flow` receives w*h interleaved (u, v) pairs, the layout
 * nlk_dev_warp_bicubic and the .flo files use. I1(x + flow(x)) ~ I0(x). */
struct nlk_tvl1_params {
  float tau;      /* time step (0.25) */
  float lambda;   /* data attachment weight (0.15) */
  float theta;    /* tightness (0.3) */
  int nscales;    /* pyramid levels actually used: cap it with nlk_tvl1_scales() */
  int fscale;     /* finest level that is solved; finer ones get the upsampled flow */
  float zfactor;  /* pyramid factor (0.5) */
  int nwarps;     /* warps per level (5) */
  float epsilon;  /* stop when the mean squared update <= epsilon^2 (0.01) */
};
void nlk_tvl1_default_params(struct nlk_tvl1_params *p);
/* number of levels the reference's command line derives from the image size */
int nlk_tvl1_scales(int w, int h, int nscales, float zfactor);
/* `iterations` (may be NULL) receives the total number of fixed-point iterations */
int nlk_dev_tvl1_flow(nlk_ctx *ctx, float *flow, const float *I0, const float *I1, int w, int h,
                      const struct nlk_tvl1_params *prms, int *iterations);
/* luminance .299 R + .587 G + .114 B of an interleaved image (ch >= 3), copy of channel 0 otherwise */
int nlk_dev_gray(nlk_ctx *ctx, float *gray, const float *im, int w, int h, int ch);
/* 255 where |backward-difference divergence of the flow| > th, else 0 */
int nlk_dev_occlusion_mask(nlk_ctx *ctx, float *mask, const float *flow, int w, int h, float th);

/* ---- multiscale wrapper (reference: lib/multiscale/multiscaler.cpp:21-107).
 * nlk_dev_image_dct: in-place whole-image DCT of an HWC image — forward = FFTW REDFT10 in both
 * directions divided by 4*w*h (dct_inplace), inverse = REDFT01 (idct_inplace).
 * nlk_dev_copy_block: the top-left bw x bh block of coefficients of `src` (row length sw)
 * into `dst` (row length dw): what decompose / recompose / merge_coarse do between transforms
 * (decompose.cpp:40-46, recompose.cpp:43-49, merge_coarse.cpp:37-43). */
int nlk_dev_image_dct(nlk_ctx *ctx, float *img, int w, int h, int ch, int inverse);
int nlk_dev_copy_block(nlk_ctx *ctx, float *dst, int dw, const float *src, int sw, int ch, int bw, int bh);

/* Row-strip form used by the multi-GPU driver. The images are a strip of the
 * frame (h rows) that already contains the search halo; targets are the patch
 * grid rows whose first image row is oy + j*step, j in [0, ngy). `acc` is a
 * planar accumulator of (ch+1) planes of h*w floats (ch weighted sums, then
 * the weights); it is ADDED to, so the caller zeroes it and may add the halo
 * rows received from its neighbours before normalising.
 * smoother != 0 selects the nlkalman_smooth_frame statistics and gain. */
int nlk_dev_frame_accumulate(nlk_ctx *ctx, float *acc, const float *cur,
                             const float *prev, const float *basic, int w, int h,
                             int ch, float sigma,
                             const struct nlkalman_params *prms, int oy, int ngy,
                             int smoother);
/* out[y][x][c] = acc_c / acc_w where acc_w > 1e-6, else cur (rows [y0, y1)) */
int nlk_dev_frame_normalize(nlk_ctx *ctx, float *out, const float *acc,
                            const float *cur, int w, int h, int ch, int y0, int y1);

/* The three phases of nlk_dev_frame_accumulate as separate calls, for an EXACT
 * processed-mask across row strips: every rank runs `strip_match` on its strip
 * and gets one 64-bit mark word per target (grid-relative, so strips can simply
 * be concatenated in grid-row order: an all-gather), `mask_commit` replays the
 * raster order (reference: src/nlkalman.c:597-600, 930-931) over the mark words
 * of the WHOLE patch grid, and `strip_group` processes the strip's targets with
 * its slice of the resulting active flags. `marks_out` / `active` are device
 * buffers of ngx*ngy uint64 / bytes; `reach` receives the R to hand to
 * mask_commit. strip_group uses the state strip_match left in the context. */
int nlk_dev_strip_match(nlk_ctx *ctx, const float *cur, const float *prev,
                        const float *basic, int w, int h, int ch, float sigma,
                        const struct nlkalman_params *prms, int oy, int ngy,
                        int smoother, void *marks_out, int *reach);
/* the same for the target rows [r0, r0 + rows) of the strip only (their records and mark words land at
 * their place in the strip's arrays): lets a rank match the rows that do not depend on the previous
 * frame's halo while that halo is still in flight, and the seam rows afterwards. Every call lays the
 * strip out again (the halo may have arrived in between). */
int nlk_dev_strip_match_rows(nlk_ctx *ctx, const float *cur, const float *prev,
                             const float *basic, int w, int h, int ch, float sigma,
                             const struct nlkalman_params *prms, int oy, int ngy,
                             int smoother, int r0, int rows, void *marks_out, int *reach);
/* the same, laying out only the pixel rows [lay0, lay1) of the strip (planar copies, validity row test) and
 * finishing the validity map of the rows [v0, v1) (row y needs the row tests of rows y .. y + patch - 1): first the
 * own rows and the targets that read nothing else, then - once the previous frame's halo rows have arrived - those
 * rows and the seam targets. Nothing is read while in flight, nothing is laid out twice. */
int nlk_dev_strip_match_part(nlk_ctx *ctx, const float *cur, const float *prev,
                             const float *basic, int w, int h, int ch, float sigma,
                             const struct nlkalman_params *prms, int oy, int ngy,
                             int smoother, int r0, int rows, int lay0, int lay1, int v0, int v1,
                             void *marks_out, int *reach);
/* Strip calls: a planar (ch+1, h, w) accumulator of the strip whose rows nlk_dev_strip_match* clear while they
 * lay the same pixel rows out (all of them over a strip's calls) - saves the caller a separate clear per step.
 * NULL (the default) switches it off. */
int nlk_ctx_set_strip_accumulator(nlk_ctx *ctx, float *acc);
int nlk_dev_mask_commit(nlk_ctx *ctx, const void *marks, int ngx, int ngy, int reach,
                        unsigned char *active);
int nlk_dev_strip_group(nlk_ctx *ctx, float *acc, const unsigned char *active);
/* nlk_dev_mask_commit over the whole grid (`marks`, `active`: ngx * ngy entries) + nlk_dev_strip_group on the strip
 * whose first grid row is gy0, as one call: where the group kernel can replay the mask inside its own launch
 * (8 x 8 patches, reach <= 3, grids up to 2048 targets wide; NLK_NO_CHASE=1 switches it off) only the grid rows down
 * to the strip's last one are replayed, by the launch's first workgroup, while the others already work (`active` is
 * then not written); otherwise exactly the two calls. Same decisions, same sums (reference: src/nlkalman.c:597-600,
 * 930-931). */
int nlk_dev_strip_commit_group(nlk_ctx *ctx, float *acc, const void *marks, int ngx, int ngy, int reach, int gy0,
                               unsigned char *active);
/* After a call whose group kernel replayed the mask itself (whole-frame calls of 8 x 8 patches,
 * nlk_dev_strip_commit_group): write the decision bytes of the replayed grid rows - all of them for a frame call
 * (into the context's records, what nlk_ctx_read_records does first), rows [0, gy0 + the strip's rows) of `active`
 * for a strip - and wait. No-op otherwise. */
int nlk_ctx_flush_active(nlk_ctx *ctx);

/* per-target records of the last frame call, copied to host (tests only):
 * active[ngrid] (1 = processed), nsel/np0/nagg[ngrid], topk[ngrid*kmax] and
 * gcoords[ngrid*gmax] packed as x | y << 16. Any pointer may be NULL. */
int nlk_ctx_read_records(nlk_ctx *ctx, int *ngrid, int *kmax, int *gmax,
                         unsigned char *active, int *nsel, int *np0, int *nagg,
                         unsigned int *topk, unsigned int *gcoords);

/* ---- One frame over several GPUs, driven from C (csrc/strips.hip; the reference's analogue is
 * the static row split of its OpenMP loop, src/nlkalman.c:586). The patch-grid rows are cut into `world` strips.
 * Per step and strip: the previous frame's halo rows come from the neighbours, the strip is matched (its interior
 * while the halo travels), every strip's 64-bit mark words go to every strip, the raster-order mask is replayed
 * over the WHOLE grid (so the output is the serial order's for any number of strips), the strip's groups are
 * filtered, the accumulator rows written outside the own rows go to their owner, the own rows are normalised.
 * Transports: RCCL over xGMI with one strip per process (grouped ncclSend / ncclRecv between neighbours, the mark
 * words as one group of ncclBroadcast; librccl is dlopen'ed - the copy the process already holds, if any), or
 * device copies with every strip in one process (`devices` may repeat an index: the whole decomposition on one
 * GPU). A step only enqueues work; nlk_strips_sync waits for it. */
typedef struct nlk_strips nlk_strips;
/* `nlocal` strips of this process = ranks rank0 .. rank0 + nlocal - 1 of `world` (nlocal == 1, or == world),
 * strip i on HIP device devices[i]. have_prev = 0: first-frame calls (no previous frame, nothing to exchange). */
int nlk_strips_create(nlk_strips **out, int nlocal, const int *devices, int rank0, int world, int w, int h,
                      int ch, float sigma, const struct nlkalman_params *prms, int smoother, int have_prev);
void nlk_strips_destroy(nlk_strips *s);
const char *nlk_strips_last_error(const nlk_strips *s);
/* one strip per process: a 128-byte communicator id made on one rank (carried to the others by whatever
 * started them), then the communicator of the world on every rank */
int nlk_rccl_unique_id(void *id128);
int nlk_strips_rccl_init(nlk_strips *s, const void *id128);
const char *nlk_strips_transport(const nlk_strips *s);
/* rows of full device-resident HWC frames (on that strip's device) into local strip `local`: cur with its halo,
 * prev with its OWN rows only (may be NULL) */
int nlk_strips_load(nlk_strips *s, int local, const float *cur_full, const float *prev_full);
/* overlap: match the interior rows while the halo travels (default 0: the two extra rounds of matching launches
 * cost more than the exchange they hide at 1080p); timing: per-phase device times, one
 * synchronisation per step (diagnosis); graph: capture the step into a HIP graph once and replay it (one strip
 * per process; falls back to plain launches by itself if the capture is refused) */
int nlk_strips_set_options(nlk_strips *s, int overlap, int timing, int graph);
/* a model, not a result: one rank of a larger world stepped ALONE with every exchange skipped (the output means
 * nothing) - what its kernels and launch gaps cost at that world size on a box with one GPU */
int nlk_strips_set_dry_run(nlk_strips *s, int on);
int nlk_strips_step(nlk_strips *s);
int nlk_strips_sync(nlk_strips *s);
/* own rows [*y0, *y1) of the output of local strip `local` (device pointer, valid until the next step); the
 * whole-grid mark words, and the decisions it used: one byte per target of the grid rows from 0 down to the strip's
 * own last row (a strip needs no later ones: the whole grid for the last strip; the rows after them are not
 * defined). Any pointer may be NULL. Call after nlk_strips_sync. */
int nlk_strips_own_rows(nlk_strips *s, int local, int *y0, int *y1, float **rows, void **marks_full,
                        unsigned char **active_full);
nlk_ctx *nlk_strips_ctx(nlk_strips *s, int local);
/* gy0, gy1 (patch-grid rows), Y0, Y1 (pixel rows held), own0, own1 (pixel rows owned) of a local strip */
int nlk_strips_geometry(nlk_strips *s, int local, int geom[6]);
/* phase_ms[7]: mean device time of [previous-frame halo, matching, mark words, mask replay, groups, accumulator
 * halos, normalisation] on local strip 0 over the steps made with timing on; *issue_us: mean host time a step
 * took to enqueue since the last call; *graph: the steps are replayed from a captured graph */
int nlk_strips_stats(nlk_strips *s, float phase_ms[7], float *issue_us, int *graph);

/* the host-side tables a frame call uploads (tests only; no device needed): the orthonormal
 * DCT-II basis [psz][psz] that stands for FFTW REDFT10/REDFT01 x the reference's scaling
 * (src/nlkalman.c:204-220, 281-298, 335-353), the aggregation window (:365-419), and the 12 x 12 matrix the
 * 12-point flow graph of the packed-lane kernel applies (csrc/k_dct12.h, evaluated on the host: column j =
 * graph(e_j)). Any pointer may be NULL. */
int nlk_host_tables(int psz, float *basis, float *window, float *basis12_regs);

#ifdef __cplusplus
}
#endif
#endif /* NLK_HIP_H */
