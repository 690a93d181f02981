/* imgio.h — frame / flow / mask file I/O of the command-line tools.
 *
 * Replaces the reference CLI's use of lib/iio (reference: src/main-flt.c:216-332,
 * 379, 387; lib/iio/iio.c:3713 read, :4501 write) with a dependency-free subset
 * that covers what the pipelines exchange:
 *   read : .tif/.tiff (classic + BigTIFF, strips or tiles, chunky or planar, 8/16/32-bit
 *          uint/int, 32/64-bit float, uncompressed / LZW / PackBits / Deflate, predictor 1/2),
 *          .pfm, .flo, .png
 *          (8/16-bit gray, gray+alpha, RGB, RGBA, palette; non-interlaced)
 *   write: by extension — .tif/.tiff (float32, or 8-bit when every sample is an
 *          integer in [0,255] like the reference's writer, lib/iio/iio.c:4310-4321;
 *          uncompressed, or LZW with NLK_TIFF_LZW=1 below 4 Mpixel like lib/iio/iio.c:3022-3026),
 *          .pfm, .flo, .png (8-bit)
 * Images are HWC interleaved float32, samples are cast without scaling. */
#ifndef NLK_IMGIO_H
#define NLK_IMGIO_H

/* returns a malloc'd w*h*ch float array or NULL (message on stderr) */
float *img_read(const char *path, int *w, int *h, int *ch);
/* returns 0 on success */
int img_write(const char *path, const float *data, int w, int h, int ch);

#endif
