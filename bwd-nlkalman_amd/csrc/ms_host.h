// ms_host.h — host side of the multiscale entry points of include/nlk_hip.h (kernels: k_ms.h)
#pragma once

namespace {

int ms_gemm(nlk_ctx* c, const float* A, long am, long ak, const float* B, long bk, long bn, float* C, long cm,
            long cn, int M, int N, int K) {
  hipLaunchKernelGGL(k_ms_gemm, dim3((N + NLK_MS_T - 1) / NLK_MS_T, (M + NLK_MS_T - 1) / NLK_MS_T), dim3(256), 0, c->stream, A, am, ak, B, bk, bn,
                     C, cm, cn, M, N, K);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

}  // namespace

extern "C" {

int nlk_dev_image_dct(nlk_ctx* c, float* img, int w, int h, int ch, int inverse) {
  if (!c || !img || w < 1 || h < 1 || ch < 1) return fail(c, NLK_EINVAL, "nlk_dev_image_dct: bad argument");
  NLK_USE_DEVICE(c);
  const size_t n = (size_t)w * h * ch;
  int rc = reserve(c, c->ms, sizeof(float) * (n + (size_t)h * h + (size_t)w * w));
  if (rc) return rc;
  float* tmp = (float*)c->ms.p;
  float* Mh = tmp + n;
  float* Mw = Mh + (size_t)h * h;
  hipLaunchKernelGGL(k_ms_basis, dim3((h + 255) / 256, h), dim3(256), 0, c->stream, Mh, h, inverse);
  hipLaunchKernelGGL(k_ms_basis, dim3((w + 255) / 256, w), dim3(256), 0, c->stream, Mw, w, inverse);
  // rows: tmp[k][x*ch+c] = sum_y Mh[k][y] img[y][x*ch+c]
  if ((rc = ms_gemm(c, Mh, h, 1, img, (long)w * ch, 1, tmp, (long)w * ch, 1, h, w * ch, h))) return rc;
  // columns, per channel: img[k][l][c] = sum_x tmp[k][x][c] Mw[l][x]
  for (int cc = 0; cc < ch; ++cc)
    if ((rc = ms_gemm(c, tmp + cc, (long)w * ch, ch, Mw, 1, w, img + cc, (long)w * ch, ch, h, w, w))) return rc;
  return NLK_OK;
}

int nlk_dev_copy_block(nlk_ctx* c, float* dst, int dw, const float* src, int sw, int ch, int bw, int bh) {
  if (!c || !dst || !src || bw < 0 || bh < 0 || bw > dw || bw > sw || ch < 1)
    return fail(c, NLK_EINVAL, "nlk_dev_copy_block: bad argument");
  if (bw == 0 || bh == 0) return NLK_OK;
  NLK_USE_DEVICE(c);
  hipLaunchKernelGGL(k_ms_copy_block, dim3((bw * ch + 255) / 256, bh), dim3(256), 0, c->stream, dst, dw, src, sw,
                     ch, bw, bh);
  HIPCHK(c, hipGetLastError());
  return NLK_OK;
}

}  // extern "C"
