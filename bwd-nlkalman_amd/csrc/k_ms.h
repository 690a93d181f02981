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
// Workgroup = 4 wavefronts computing a 64 x 64 tile, each wavefront 32 x 32 as 2 x 2 MFMA tiles
// of v_mfma_f32_16x16x4_f32; K is walked in chunks of 16 staged through LDS.
typedef float nlk_ms_f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_ms_gemm(const float* __restrict__ A, long am, long ak, const float* __restrict__ B, long bk, long bn,
          float* __restrict__ C, long cm, long cn, int M, int N, int K) {
  __shared__ float As[16][64 + 4], Bs[16][64 + 4];  // [k][m], [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  nlk_ms_f4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = nlk_ms_f4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
    // stage: 1024 elements of each operand, 4 per thread; consecutive threads walk the
    // operand's fastest dimension
    for (int e = tid; e < 1024; e += 256) {
      {  // A tile: element (m, k); consecutive threads follow the smaller stride
        const int kk = ak <= am ? (e & 15) : (e >> 6), mm = ak <= am ? (e >> 4) : (e & 63);
        const int m = m0 + mm, k = k0 + kk;
        As[kk][mm] = (m < M && k < K) ? A[m * am + k * ak] : 0.f;
      }
      {  // B tile: element (k, n)
        const int kk = bn <= bk ? (e >> 6) : (e & 15), nn = bn <= bk ? (e & 63) : (e >> 4);
        const int k = k0 + kk, n = n0 + nn;
        Bs[kk][nn] = (k < K && n < N) ? B[k * bk + n * bn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[ks * 4 + (lane >> 4)][wm + t * 16 + (lane & 15)];
        b[t] = Bs[ks * 4 + (lane >> 4)][wn + t * 16 + (lane & 15)];
      }
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout: lane l, register r -> row 4*(l>>4) + r, column l & 15
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
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
