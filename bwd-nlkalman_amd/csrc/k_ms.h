// k_ms.h — whole-image DCT of the multiscale wrapper (reference:
// lib/multiscale/multiscaler.cpp:21-107: FFTW REDFT10 in both directions divided by
// 4*rows*cols, inverse = plain REDFT01), as two dense matrix products per channel on the f32
// matrix cores: Out = M_h * X * M_w^T with
//   forward  M[k][j] = cos(pi (j + 1/2) k / n) / n          (k = frequency, j = sample)
//   inverse  M[k][j] = j == 0 ? 1 : 2 cos(pi j (k + 1/2) / n)  (k = sample, j = frequency)
// The image stays HWC interleaved: the first product contracts over the rows (its columns are
// the w*ch interleaved samples), the second one runs per channel with an element stride of ch.
#pragma once
#include <hip/hip_runtime.h>

__global__ void k_ms_basis(float* __restrict__ M, int n, int inverse) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (j >= n) return;
  const double pi = 3.14159265358979323846;
  double v;
  if (!inverse) v = cos(pi * (j + 0.5) * k / n) / n;
  else v = j == 0 ? 1.0 : 2.0 * cos(pi * j * (k + 0.5) / n);
  M[(size_t)k * n + j] = (float)v;
}

// C[m*cm + n*cn] = sum_k A[m*am + k*ak] * B[k*bk + n*bn]   (general strides, f32)
// Workgroup = 4 wavefronts computing a 128 x 128 tile, each wavefront 64 x 64 as 4 x 4 tiles of
// v_mfma_f32_16x16x4_f32 (16 MFMAs per 8 LDS operand reads); K is walked in chunks of 16
// staged through LDS, the next chunk's global loads in flight while the current one is
// multiplied.
typedef float nlk_ms_f4 __attribute__((ext_vector_type(4)));
#define NLK_MS_T 128  // tile edge
#define NLK_MS_KC 16  // K chunk

__global__ void __launch_bounds__(256)
k_ms_gemm(const float* __restrict__ A, long am, long ak, const float* __restrict__ B, long bk, long bn,
          float* __restrict__ C, long cm, long cn, int M, int N, int K) {
  __shared__ float As[NLK_MS_KC][NLK_MS_T + 4], Bs[NLK_MS_KC][NLK_MS_T + 4];  // [k][m], [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * NLK_MS_T, n0 = blockIdx.x * NLK_MS_T;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  nlk_ms_f4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = nlk_ms_f4{0.f, 0.f, 0.f, 0.f};
  // staging roles: 2048 elements of each operand per chunk, 8 per thread; consecutive threads
  // follow the operand's smaller stride
  constexpr int PER = NLK_MS_T * NLK_MS_KC / 256;
  const bool a_kfast = ak <= am, b_nfast = bn <= bk;
  float ra[PER], rb[PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u;
      {
        const int kk = a_kfast ? (e & 15) : (e >> 7), mm = a_kfast ? (e >> 4) : (e & 127);
        const int m = m0 + mm, k = k0 + kk;
        ra[u] = (m < M && k < K) ? A[m * am + k * ak] : 0.f;
      }
      {
        const int kk = b_nfast ? (e >> 7) : (e & 15), nn = b_nfast ? (e & 127) : (e >> 4);
        const int k = k0 + kk, n = n0 + nn;
        rb[u] = (k < K && n < N) ? B[k * bk + n * bn] : 0.f;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u;
      As[a_kfast ? (e & 15) : (e >> 7)][a_kfast ? (e >> 4) : (e & 127)] = ra[u];
      Bs[b_nfast ? (e >> 7) : (e & 15)][b_nfast ? (e & 127) : (e >> 4)] = rb[u];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += NLK_MS_KC) {
    stage();
    __syncthreads();
    if (k0 + NLK_MS_KC < K) fetch(k0 + NLK_MS_KC);
#pragma unroll
    for (int ks = 0; ks < NLK_MS_KC / 4; ++ks) {
      float a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = As[ks * 4 + (lane >> 4)][wm + t * 16 + (lane & 15)];
        b[t] = Bs[ks * 4 + (lane >> 4)][wn + t * 16 + (lane & 15)];
      }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout: lane l, register r -> row 4*(l>>4) + r, column l & 15
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + ta * 16 + 4 * (lane >> 4) + r, n = n0 + wn + tb * 16 + (lane & 15);
        if (m < M && n < N) C[m * cm + n * cn] = acc[ta][tb][r];
      }
}

// dst[y][x][c] = src[y][x][c] for y < bh, x < bw (top-left block of coefficients), images of
// different widths (reference: lib/multiscale/decompose.cpp:40-46, recompose.cpp:43-49)
__global__ void k_ms_copy_block(float* __restrict__ dst, int dw, const float* __restrict__ src, int sw,
                                int ch, int bw, int bh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (i >= bw * ch || y >= bh) return;
  dst[((size_t)y * dw) * ch + i] = src[((size_t)y * sw) * ch + i];
}
