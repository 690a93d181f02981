/* nlk_oracle.h — interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see the
 * header of nlk_oracle.c: parity unpinned). */
#ifndef NLK_ORACLE_H
#define NLK_ORACLE_H
#include <stdint.h>

enum { NLKO_FLT1 = 0, NLKO_FLT2 = 1, NLKO_SMO1 = 2 };

/* same field order as struct nlkalman_params (reference: src/nlkalman.h:22-37) */
typedef struct {
  int patch_sz, search_sz_x, search_sz_t, npatches_x, npatches_t, npatches_tagg;
  float dista_lambda, beta_x, beta_t;
} nlko_params;

/* optional per-target trace, indexed by grid index gx + gy*ngx; every pointer
 * may be NULL. Coordinates are packed as x | y << 16, -1 = unused entry. */
typedef struct {
  int kmax;      /* row length of topk */
  int gmax;      /* row length of gcoords */
  int *topk;     /* [ngrid][kmax] kept candidates in sorted order */
  int *gcoords;  /* [ngrid][gmax] aggregated group members */
  int *nsel, *np0, *np1, *nagg, *active;
  float *vp;
  float *aggr;   /* [w*h] aggregation weights before normalisation */
} nlko_trace;

void nlko_default_params(nlko_params *p, float sigma, int mode);
void nlko_window(float *W, int psz);
void nlko_rgb2opp(float *im, int w, int h, int ch);
void nlko_opp2rgb(float *im, int w, int h, int ch);
void nlko_warp_bicubic(float *imw, const float *im, const float *of,
                       const float *msk, int w, int h, int ch);
void nlko_awgn(float *x, long n, float sigma, uint32_t seed);
void nlko_dct_basis(float *C, int n);
void nlko_dct2(float *planes, int n, int nplanes, int inverse);
void nlko_filter_frame(float *deno1, const float *nisy1, const float *deno0,
                       const float *bsic1, int w, int h, int ch, float sigma,
                       const nlko_params *P, int nthreads, nlko_trace *tr);
void nlko_smooth_frame(float *smoo1, const float *filt1, const float *smoo0,
                       const float *bsic1, int w, int h, int ch, float sigma,
                       const nlko_params *P, int nthreads, nlko_trace *tr);
void nlko_frame_accumulate(float *acc, const float *cur, const float *prev,
                           const float *basic, int w, int h, int ch, float sigma,
                           const nlko_params *P, int oy, int ngy, int smoother);
void nlko_frame_normalize(float *out, const float *acc, const float *cur, int w, int h,
                          int ch, int y0, int y1);
void nlko_strip_match(uint64_t *marks, const float *cur, const float *prev,
                      const float *basic, int w, int h, int ch, float sigma,
                      const nlko_params *P, int oy, int ngy, int smoother, int *reach);
void nlko_mask_commit(const uint64_t *marks, int ngx, int ngy, int R, unsigned char *active);
void nlko_strip_group(float *acc, const unsigned char *active, const float *cur,
                      const float *prev, const float *basic, int w, int h, int ch,
                      float sigma, const nlko_params *P, int oy, int ngy, int smoother);
int nlko_max_threads(void);

/* ---- dual TV-L1 optical flow + occlusion mask (tvl1_oracle.c; reference: lib/tvl1flow/) */
float tvl1o_bicubic_at(const float *im, float uu, float vv, int nx, int ny, int border_out);
void tvl1o_forward_gradient(const float *f, float *fx, float *fy, int nx, int ny);
void tvl1o_centered_gradient(const float *f, float *dx, float *dy, int nx, int ny);
void tvl1o_divergence(const float *v1, const float *v2, float *div, int nx, int ny);
void tvl1o_gaussian(float *im, int nx, int ny, double sigma);
void tvl1o_zoom_size(int nx, int ny, int *nxx, int *nyy, float factor);
void tvl1o_zoom_out(const float *im, float *out, int nx, int ny, float factor);
void tvl1o_zoom_in(const float *im, float *out, int nx, int ny, int nxx, int nyy);
int tvl1o_flow_scale(const float *I0, const float *I1, float *u1, float *u2, int nx, int ny,
                     float tau, float lambda, float theta, int warps, float epsilon, int *iters_out);
void tvl1o_normalize(const float *I0, const float *I1, float *o0, float *o1, int n);
int tvl1o_auto_scales(int nx, int ny, int nscales, float zfactor);
void tvl1o_flow(const float *I0, const float *I1, float *u1, float *u2, int nx, int ny, float tau,
                float lambda, float theta, int nscales, int fscale, float zfactor, int warps,
                float epsilon);
void tvl1o_occlusion_mask(const float *flow, float *mask, int nx, int ny, float th);

/* ---- multiscale wrapper (ms_oracle.c; reference: lib/multiscale/multiscaler.cpp) */
void mso_image_dct(float *img, int w, int h, int ch, int inverse);
#endif
