/* imgio_jpeg.c — baseline JPEG input for imgio.c (the reference's library reads JPEG through libjpeg,
 * lib/iio/iio.c:1416-1460: 8-bit samples, one grey or three RGB channels, values 0..255 as floats).
 *
 * Own code from the format specification (ITU T.81: sequential DCT, Huffman coding, 8-bit precision, restart
 * intervals, any sampling factors up to 4; JFIF: three components are Y Cb Cr). Not read: progressive, arithmetic
 * coded, lossless and 12-bit files, four components (CMYK / YCCK).
 *
 * What has to agree with libjpeg to give the same NUMBERS, not just the same picture:
 *  - the inverse transform of a full-resolution component is the integer "slow but accurate" 8 x 8 algorithm every
 *    libjpeg uses by default (Loeffler-Ligtenberg-Moschytz, 13-bit constants, two passes with 2 extra bits between);
 *  - Y Cb Cr -> R G B with the 16-bit fixed-point tables of its colour converter (1.402, 0.344136286, 0.714136286,
 *    1.772);
 *  - a subsampled chroma component is brought to full resolution the way libjpeg 7 and later do it: by the inverse
 *    transform itself, which produces 16 samples from the 8 coefficients of a block (a 16-point transform in the same
 *    13-bit fixed point), not by interpolating samples; factors the transform cannot take (3, and the second half of
 *    a 4) are plain replication.
 * tests/test_reference_tools.py compares against the reference's library: bit for bit (grey, 4:4:4, 4:2:2, 4:2:0). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

float *nlk_read_jpeg(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch);

static float *jfail(const char *path, const char *why) {
  fprintf(stderr, "imgio: %s: %s\n", path, why);
  return NULL;
}

static const unsigned char ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct huff {
  int present;
  unsigned char counts[17], symbols[256];
  int32_t mincode[17], maxcode[18], valptr[17];
};

static void huff_build(struct huff *t) {
  int32_t code = 0;
  int k = 0;
  for (int l = 1; l <= 16; ++l) {
    t->valptr[l] = k;
    t->mincode[l] = code;
    code += t->counts[l];
    k += t->counts[l];
    t->maxcode[l] = t->counts[l] ? code - 1 : -1;
    code <<= 1;
  }
  t->maxcode[17] = 0x7fffffff;
}

struct bits {
  const unsigned char *p, *end;
  uint32_t acc;
  int cnt;
  int marker; /* a marker met inside the entropy-coded data (restart or the end) */
};

static int bits_fill(struct bits *s, int need) {
  while (s->cnt < need) {
    int c = 0;
    if (!s->marker && s->p < s->end) {
      c = *s->p++;
      if (c == 0xFF) {
        int d = s->p < s->end ? *s->p : 0xD9;
        if (d == 0) ++s->p;              /* a stuffed zero: the data byte FF */
        else { s->marker = d; --s->p; c = 0; } /* zeros from here on, like every decoder feeds them */
      }
    }
    s->acc = (s->acc << 8) | (uint32_t)c;
    s->cnt += 8;
  }
  return 0;
}

static int bits_get(struct bits *s, int nb) {
  if (!nb) return 0;
  bits_fill(s, nb);
  s->cnt -= nb;
  return (int)((s->acc >> s->cnt) & ((1u << nb) - 1u));
}

static int huff_decode(struct bits *s, const struct huff *t) {
  int32_t code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | bits_get(s, 1);
    if (t->maxcode[l] >= 0 && code <= t->maxcode[l] && code >= t->mincode[l])
      return t->symbols[t->valptr[l] + (code - t->mincode[l])];
  }
  return -1;
}

static int extend(int v, int nb) { return nb && v < (1 << (nb - 1)) ? v - (1 << nb) + 1 : v; }

/* ---- the 8 x 8 inverse transform in integers (13-bit constants, 2 extra bits after the column pass) */
#define CB 13
#define P1 2
#define F_0_298631336 2446
#define F_0_390180644 3196
#define F_0_541196100 4433
#define F_0_765366865 6270
#define F_0_899976223 7373
#define F_1_175875602 9633
#define F_1_501321110 12299
#define F_1_847759065 15137
#define F_1_961570560 16069
#define F_2_053119869 16819
#define F_2_562915447 20995
#define F_3_072711026 25172

static void idct1d(const int32_t in[8], int32_t out[8], int32_t round, int shift) {
  int32_t z1 = (in[2] + in[6]) * F_0_541196100;
  const int32_t t2 = z1 + in[2] * F_0_765366865, t3 = z1 - in[6] * F_1_847759065;
  const int32_t e0 = (int32_t)((uint32_t)in[0] << CB) + round, e4 = (int32_t)((uint32_t)in[4] << CB);
  const int32_t t0 = e0 + e4, t1 = e0 - e4;
  const int32_t t10 = t0 + t2, t13 = t0 - t2, t11 = t1 + t3, t12 = t1 - t3;
  int32_t o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
  int32_t z2 = o0 + o2, z3 = o1 + o3;
  z1 = (z2 + z3) * F_1_175875602;
  z2 = z2 * -F_1_961570560 + z1;
  z3 = z3 * -F_0_390180644 + z1;
  z1 = (o0 + o3) * -F_0_899976223;
  o0 = o0 * F_0_298631336 + z1 + z2;
  o3 = o3 * F_1_501321110 + z1 + z3;
  z1 = (o1 + o2) * -F_2_562915447;
  o1 = o1 * F_2_053119869 + z1 + z3;
  o2 = o2 * F_3_072711026 + z1 + z2;
  out[0] = (t10 + o3) >> shift; out[7] = (t10 - o3) >> shift;
  out[1] = (t11 + o2) >> shift; out[6] = (t11 - o2) >> shift;
  out[2] = (t12 + o1) >> shift; out[5] = (t12 - o1) >> shift;
  out[3] = (t13 + o0) >> shift; out[4] = (t13 - o0) >> shift;
}

static unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static void idct8x8(const int32_t *coef, unsigned char *dst, size_t stride) {
  int32_t ws[64], col[8], res[8];
  for (int x = 0; x < 8; ++x) { /* columns */
    for (int y = 0; y < 8; ++y) col[y] = coef[8 * y + x];
    idct1d(col, res, 1 << (CB - P1 - 1), CB - P1);
    for (int y = 0; y < 8; ++y) ws[8 * y + x] = res[y];
  }
  for (int y = 0; y < 8; ++y) { /* rows; + 128 and the rounding of the last shift in the first term */
    idct1d(ws + 8 * y, res, (128 << (CB + P1 + 3)) + (1 << (CB + P1 + 2)), CB + P1 + 3);
    for (int x = 0; x < 8; ++x) dst[y * stride + x] = clamp8(res[x]);
  }
}

/* ---- 16 output samples from the 8 coefficients of a block: the larger inverse transform libjpeg (7 and later)
 * brings subsampled chroma to full resolution with, in the same fixed point as the 8-point one. cK = sqrt(2)
 * cos(K pi / 32); the constants are those combinations rounded to 13 bits. */
#define FX(x) ((int32_t)((x) * (1 << CB) + 0.5))
static void idct1d_16(const int32_t in[8], int32_t out[16], int32_t round, int shift) {
  int32_t t0 = (int32_t)((uint32_t)in[0] << CB) + round;
  int32_t z1 = in[4];
  int32_t t1 = z1 * FX(1.306562965), t2 = z1 * F_0_541196100;
  const int32_t t10 = t0 + t1, t11 = t0 - t1, t12 = t0 + t2, t13 = t0 - t2;
  z1 = in[2];
  int32_t z2 = in[6], z3 = z1 - z2;
  int32_t z4 = z3 * FX(0.275899379);
  z3 = z3 * FX(1.387039845);
  t0 = z3 + z2 * F_2_562915447;
  t1 = z4 + z1 * F_0_899976223;
  t2 = z3 - z1 * FX(0.601344887);
  int32_t t3 = z4 - z2 * FX(0.509795579);
  const int32_t t20 = t10 + t0, t27 = t10 - t0, t21 = t12 + t1, t26 = t12 - t1;
  const int32_t t22 = t13 + t2, t25 = t13 - t2, t23 = t11 + t3, t24 = t11 - t3;
  z1 = in[1]; z2 = in[3]; z3 = in[5]; z4 = in[7];
  int32_t o11 = z1 + z3;
  int32_t o1 = (z1 + z2) * FX(1.353318001), o2 = o11 * FX(1.247225013), o3 = (z1 + z4) * FX(1.093201867);
  int32_t o10 = (z1 - z4) * FX(0.897167586);
  o11 = o11 * FX(0.666655658);
  int32_t o12 = (z1 - z2) * FX(0.410524528);
  const int32_t o0 = o1 + o2 + o3 - z1 * FX(2.286341144);
  const int32_t o13 = o10 + o11 + o12 - z1 * FX(1.835730603);
  z1 = (z2 + z3) * FX(0.138617169);
  o1 += z1 + z2 * FX(0.071888074);
  o2 += z1 - z3 * FX(1.125726048);
  z1 = (z3 - z2) * FX(1.407403738);
  o11 += z1 - z3 * FX(0.766367282);
  o12 += z1 + z2 * FX(1.971951411);
  z2 += z4;
  z1 = z2 * -FX(0.666655658);
  o1 += z1;
  o3 += z1 + z4 * FX(1.065388962);
  z2 = z2 * -FX(1.247225013);
  o10 += z2 + z4 * FX(3.141271809);
  o12 += z2;
  z2 = (z3 + z4) * -FX(1.353318001);
  o2 += z2;
  o3 += z2;
  z2 = (z4 - z3) * FX(0.410524528);
  o10 += z2;
  o11 += z2;
  out[0] = (t20 + o0) >> shift;   out[15] = (t20 - o0) >> shift;
  out[1] = (t21 + o1) >> shift;   out[14] = (t21 - o1) >> shift;
  out[2] = (t22 + o2) >> shift;   out[13] = (t22 - o2) >> shift;
  out[3] = (t23 + o3) >> shift;   out[12] = (t23 - o3) >> shift;
  out[4] = (t24 + o10) >> shift;  out[11] = (t24 - o10) >> shift;
  out[5] = (t25 + o11) >> shift;  out[10] = (t25 - o11) >> shift;
  out[6] = (t26 + o12) >> shift;  out[9] = (t26 - o12) >> shift;
  out[7] = (t27 + o13) >> shift;  out[8] = (t27 - o13) >> shift;
}

/* a block at twice the resolution along x, y or both, in integers */
static void idct_double(const int32_t *coef, int fx, int fy, unsigned char *dst, size_t stride) {
  const int ny = 8 * fy, nx = 8 * fx;
  int32_t ws[16 * 8], col[8], res[16];
  for (int x = 0; x < 8; ++x) { /* columns: 8 or 16 rows of work space */
    for (int y = 0; y < 8; ++y) col[y] = coef[8 * y + x];
    if (fy == 2) idct1d_16(col, res, 1 << (CB - P1 - 1), CB - P1);
    else idct1d(col, res, 1 << (CB - P1 - 1), CB - P1);
    for (int y = 0; y < ny; ++y) ws[8 * y + x] = res[y];
  }
  for (int y = 0; y < ny; ++y) {
    if (fx == 2) idct1d_16(ws + 8 * y, res, (128 << (CB + P1 + 3)) + (1 << (CB + P1 + 2)), CB + P1 + 3);
    else idct1d(ws + 8 * y, res, (128 << (CB + P1 + 3)) + (1 << (CB + P1 + 2)), CB + P1 + 3);
    for (int x = 0; x < nx; ++x) dst[y * stride + x] = clamp8(res[x]);
  }
}

/* ---- a block of a component subsampled fx x fy times, at full resolution: the transform doubles what it can (one
 * doubling per direction, where the factor is even), plain replication does the rest - libjpeg's order of things */
static void block_to_full(const int32_t *coef, int fx, int fy, unsigned char *dst, size_t stride) {
  const int dx = fx % 2 == 0 ? 2 : 1, dy = fy % 2 == 0 ? 2 : 1, rx = fx / dx, ry = fy / dy;
  if (rx == 1 && ry == 1) {
    idct_double(coef, dx, dy, dst, stride);
    return;
  }
  unsigned char tmp[16 * 16];
  idct_double(coef, dx, dy, tmp, 16);
  for (int y = 0; y < 8 * fy; ++y)
    for (int x = 0; x < 8 * fx; ++x) dst[y * stride + x] = tmp[(y / ry) * 16 + x / rx];
}

struct comp {
  int id, hs, vs, tq, td, ta;
  int bw, bh;      /* blocks per row / column, padded to whole MCUs */
  int32_t *coef;   /* [bh][bw][64], dequantised */
  int pred;
  unsigned char *pix; /* full resolution, padded */
  size_t stride;
};

float *nlk_read_jpeg(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  uint32_t qt[4][64];
  int have_qt[4] = {0, 0, 0, 0};
  struct huff dc[4], ac[4];
  struct comp C[3];
  memset(dc, 0, sizeof dc);
  memset(ac, 0, sizeof ac);
  memset(C, 0, sizeof C);
  int nc = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, have_sof = 0, decoded = 0, plain_rgb = 0;
  float *out = NULL;
  size_t i = 2;
  while (i + 4 <= n && !decoded) {
    if (b[i] != 0xFF) { ++i; continue; }
    const int m = b[i + 1];
    if (m == 0xFF) { ++i; continue; }
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
    if (m == 0xD9) break;
    const size_t len = (size_t)b[i + 2] << 8 | b[i + 3];
    if (len < 2 || i + 2 + len > n) return jfail(path, "truncated JPEG segment");
    const unsigned char *s = b + i + 4, *e = b + i + 2 + len;
    i += 2 + len;
    if (m == 0xDB) { /* quantisation tables */
      while (s < e) {
        const int pq = s[0] >> 4, tq = s[0] & 15;
        if (tq > 3 || pq > 1 || s + 1 + 64 * (pq + 1) > e) return jfail(path, "bad JPEG quantisation table");
        ++s;
        for (int k = 0; k < 64; ++k, s += pq + 1) qt[tq][ZIGZAG[k]] = pq ? (uint32_t)(s[0] << 8 | s[1]) : s[0];
        have_qt[tq] = 1;
      }
    } else if (m == 0xC4) { /* Huffman tables */
      while (s < e) {
        const int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3 || s + 17 > e) return jfail(path, "bad JPEG Huffman table");
        struct huff *t = tc ? &ac[th] : &dc[th];
        int total = 0;
        t->counts[0] = 0;
        for (int l = 1; l <= 16; ++l) total += (t->counts[l] = s[l]);
        if (total > 256 || s + 17 + total > e) return jfail(path, "bad JPEG Huffman table");
        memcpy(t->symbols, s + 17, (size_t)total);
        huff_build(t);
        t->present = 1;
        s += 17 + total;
      }
    } else if (m == 0xC0 || m == 0xC1) { /* frame header: sequential DCT, Huffman */
      if (e - s < 6) return jfail(path, "bad JPEG frame header");
      if (s[0] != 8) return jfail(path, "JPEG precision other than 8 bits is not supported");
      H = s[1] << 8 | s[2];
      W = s[3] << 8 | s[4];
      nc = s[5];
      if ((nc != 1 && nc != 3) || e - s < 6 + 3 * nc) return jfail(path, "JPEG with other than 1 or 3 components is not supported");
      if (W <= 0 || H <= 0 || W > 65500 || H > 65500) return jfail(path, "bad JPEG size");
      for (int c = 0; c < nc; ++c) {
        C[c].id = s[6 + 3 * c];
        C[c].hs = s[7 + 3 * c] >> 4;
        C[c].vs = s[7 + 3 * c] & 15;
        C[c].tq = s[8 + 3 * c];
        if (C[c].hs < 1 || C[c].hs > 4 || C[c].vs < 1 || C[c].vs > 4 || C[c].tq > 3) return jfail(path, "bad JPEG sampling factors");
        if (C[c].hs > hmax) hmax = C[c].hs;
        if (C[c].vs > vmax) vmax = C[c].vs;
      }
      if (nc == 1) C[0].hs = C[0].vs = hmax = vmax = 1; /* (a single component is never interleaved) */
      for (int c = 0; c < nc; ++c)
        if (hmax % C[c].hs || vmax % C[c].vs || hmax / C[c].hs > 4 || vmax / C[c].vs > 4)
          return jfail(path, "JPEG sampling factors that do not divide each other are not supported");
      have_sof = 1;
    } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      return jfail(path, "progressive / lossless / arithmetic-coded JPEG is not supported (baseline only)");
    } else if (m == 0xEE) { /* Adobe: transform 0 = the three components ARE red, green, blue */
      if (e - s >= 12 && !memcmp(s, "Adobe", 5)) plain_rgb = s[11] == 0;
    } else if (m == 0xDD) {
      if (e - s < 2) return jfail(path, "bad JPEG restart interval");
      restart = s[0] << 8 | s[1];
    } else if (m == 0xDA) { /* the scan: all components, interleaved */
      if (!have_sof || e - s < 1 || s[0] != nc || e - s < 1 + 2 * nc + 3) return jfail(path, "JPEG scan does not hold every component");
      for (int k = 0; k < nc; ++k) {
        int c = 0;
        while (c < nc && C[c].id != s[1 + 2 * k]) ++c;
        if (c == nc) return jfail(path, "bad JPEG scan header");
        C[c].td = s[2 + 2 * k] >> 4;
        C[c].ta = s[2 + 2 * k] & 15;
        if (C[c].td > 3 || C[c].ta > 3 || !dc[C[c].td].present || !ac[C[c].ta].present || !have_qt[C[c].tq])
          return jfail(path, "JPEG scan uses a table the file does not define");
      }
      const int mcuw = 8 * hmax, mcuh = 8 * vmax, mx = (W + mcuw - 1) / mcuw, my = (H + mcuh - 1) / mcuh;
      int ok = 1;
      for (int c = 0; c < nc; ++c) {
        C[c].bw = mx * C[c].hs;
        C[c].bh = my * C[c].vs;
        C[c].coef = calloc((size_t)C[c].bw * C[c].bh * 64, sizeof(int32_t));
        C[c].stride = (size_t)mx * mcuw;
        C[c].pix = malloc(C[c].stride * (size_t)my * mcuh);
        C[c].pred = 0;
        ok = ok && C[c].coef && C[c].pix;
      }
      struct bits st = {b + i, b + n, 0, 0, 0};
      int until_restart = restart;
      for (int mcu = 0; ok && mcu < mx * my; ++mcu) {
        if (restart && until_restart == 0) { /* byte-align, pass the RSTn marker, reset the predictions */
          st.cnt = 0;
          st.acc = 0;
          if (st.marker >= 0xD0 && st.marker <= 0xD7) { st.p += 2; st.marker = 0; }
          else {
            while (st.p + 1 < st.end && !(st.p[0] == 0xFF && st.p[1] >= 0xD0 && st.p[1] <= 0xD7)) ++st.p;
            st.p += 2;
            st.marker = 0;
          }
          for (int c = 0; c < nc; ++c) C[c].pred = 0;
          until_restart = restart;
        }
        --until_restart;
        const int mxi = mcu % mx, myi = mcu / mx;
        for (int c = 0; ok && c < nc; ++c)
          for (int v = 0; ok && v < C[c].vs; ++v)
            for (int hh = 0; ok && hh < C[c].hs; ++hh) {
              int32_t *blk = C[c].coef + ((size_t)(myi * C[c].vs + v) * C[c].bw + (mxi * C[c].hs + hh)) * 64;
              const uint32_t *q = qt[C[c].tq];
              int t = huff_decode(&st, &dc[C[c].td]);
              if (t < 0 || t > 11) { ok = 0; break; }
              C[c].pred += extend(bits_get(&st, t), t);
              blk[0] = (int32_t)(C[c].pred * q[0]);
              for (int k = 1; k < 64;) {
                const int rs = huff_decode(&st, &ac[C[c].ta]);
                if (rs < 0) { ok = 0; break; }
                const int r = rs >> 4, sz = rs & 15;
                if (!sz) {
                  if (r == 15) { k += 16; continue; }
                  break; /* end of block */
                }
                k += r;
                if (k > 63) { ok = 0; break; }
                blk[ZIGZAG[k]] = (int32_t)(extend(bits_get(&st, sz), sz) * q[ZIGZAG[k]]);
                ++k;
              }
            }
      }
      if (ok) { /* samples: every component at full resolution */
        for (int c = 0; c < nc; ++c) {
          const int fx = hmax / C[c].hs, fy = vmax / C[c].vs;
          for (int by = 0; by < C[c].bh; ++by)
            for (int bx = 0; bx < C[c].bw; ++bx) {
              const int32_t *blk = C[c].coef + ((size_t)by * C[c].bw + bx) * 64;
              unsigned char *dst = C[c].pix + (size_t)by * 8 * fy * C[c].stride + (size_t)bx * 8 * fx;
              if (fx == 1 && fy == 1) idct8x8(blk, dst, C[c].stride);
              else block_to_full(blk, fx, fy, dst, C[c].stride);
            }
        }
        out = malloc((size_t)W * H * nc * sizeof(float));
        if (out) {
          for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
              float *o = out + ((size_t)y * W + x) * nc;
              const int Y = C[0].pix[(size_t)y * C[0].stride + x];
              if (nc == 1) { o[0] = (float)Y; continue; }
              if (plain_rgb) {
                o[0] = (float)Y; o[1] = C[1].pix[(size_t)y * C[1].stride + x]; o[2] = C[2].pix[(size_t)y * C[2].stride + x];
                continue;
              }
              const int cb = C[1].pix[(size_t)y * C[1].stride + x] - 128, cr = C[2].pix[(size_t)y * C[2].stride + x] - 128;
              /* 16-bit fixed point: 91881 = 1.402, 116130 = 1.772, 46802 = 0.714136286, 22553 = 0.344136286 */
              const int r = Y + ((91881 * cr + 32768) >> 16), bl = Y + ((116130 * cb + 32768) >> 16);
              const int g = Y + ((-22553 * cb + 32768 - 46802 * cr) >> 16);
              o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(bl);
            }
        }
      }
      for (int c = 0; c < nc; ++c) { free(C[c].coef); free(C[c].pix); C[c].coef = NULL; C[c].pix = NULL; }
      if (!ok) return jfail(path, "corrupt JPEG data");
      if (!out) return jfail(path, "out of memory");
      decoded = 1;
    }
  }
  if (!decoded) return jfail(path, "JPEG without a scan");
  *w = W; *h = H; *ch = nc;
  return out;
}
