/* ms_oracle.c — CPU restatement of the multiscale wrapper's whole-image DCT
 * (reference: lib/multiscale/multiscaler.cpp:21-107).
 *
 * TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).
 *
 * PARITY UNPINNED: the reference calls FFTW (REDFT10 / REDFT01, fftw3f, no pinned version),
 * which is absent from this image, and has no tests or golden vectors for it. The transform
 * is restated from FFTW's published definition
 *   REDFT10: Y_k = 2 sum_j X_j cos(pi (j + 1/2) k / n)
 *   REDFT01: Y_k = X_0 + 2 sum_{j>=1} X_j cos(pi j (k + 1/2) / n)
 * with the reference's own scaling (forward divided by 4*rows*cols, multiscaler.cpp:56-59;
 * ISOMETRIC_DCT is not defined, multiscaler.hpp:34), evaluated directly in double, and is
 * checked against scipy.fft.dctn / idctn in tests/test_multiscale.py. */
#include <math.h>
#include <stdlib.h>

#include "nlk_oracle.h"

/* 1-D transform of `count` lines of length n (element stride es, line stride ls), in place */
static void lines(float *d, int n, long es, int count, long ls, int inverse) {
  const double pi = 3.14159265358979323846;
  double *c = malloc(sizeof(double) * (size_t)n * n), *t = malloc(sizeof(double) * n);
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < n; ++j)
      c[(size_t)k * n + j] = inverse ? (j == 0 ? 1.0 : 2.0 * cos(pi * j * (k + 0.5) / n))
                                     : 2.0 * cos(pi * (j + 0.5) * k / n);
  for (int l = 0; l < count; ++l) {
    float *x = d + l * ls;
    for (int k = 0; k < n; ++k) {
      double s = 0;
      for (int j = 0; j < n; ++j) s += c[(size_t)k * n + j] * x[j * es];
      t[k] = s;
    }
    for (int k = 0; k < n; ++k) x[k * es] = (float)t[k];
  }
  free(c);
  free(t);
}

/* in-place DCT of an HWC image: forward = dct_inplace (:21-61), inverse = idct_inplace (:63-107) */
void mso_image_dct(float *img, int w, int h, int ch, int inverse) {
  /* along x: lines of length w, element stride ch; one line per (row, channel) */
  for (int y = 0; y < h; ++y) lines(img + (size_t)y * w * ch, w, ch, ch, 1, inverse);
  /* along y: lines of length h, element stride w*ch; one line per (column, channel) */
  lines(img, h, (long)w * ch, w * ch, 1, inverse);
  if (!inverse) {
    const size_t n = (size_t)w * h * ch;
    for (size_t i = 0; i < n; ++i) img[i] /= 4 * h * w;  /* :56-59 */
  }
}
