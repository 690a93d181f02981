/* imgio_jpeg.c — JPEG input (baseline and progressive) for imgio.c (the reference's library reads JPEG through libjpeg,
 * lib/iio/iio.c:1416-1460: 8-bit samples, one grey or three RGB channels, values 0..255 as floats).
 *
 * Own code from the format specification (ITU T.81: sequential and progressive DCT, Huffman coding, 8-bit precision,
 * one or several scans, restart intervals, any sampling factors up to 4; JFIF: three components are Y Cb Cr). Not
 * read: arithmetic-coded, lossless, hierarchical and 12-bit files, four components (CMYK / YCCK).
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

/* (64-bit intermediates, like libjpeg's `long`: products of extreme coefficients do not fit 32 bits) */
static void idct1d(const int32_t in[8], int32_t out[8], int64_t round, int shift) {
  int64_t z1 = ((int64_t)in[2] + in[6]) * F_0_541196100;
  const int64_t t2 = z1 + (int64_t)in[2] * F_0_765366865, t3 = z1 - (int64_t)in[6] * F_1_847759065;
  const int64_t e0 = (int64_t)in[0] * (1 << CB) + round, e4 = (int64_t)in[4] * (1 << CB);
  const int64_t t0 = e0 + e4, t1 = e0 - e4;
  const int64_t t10 = t0 + t2, t13 = t0 - t2, t11 = t1 + t3, t12 = t1 - t3;
  int64_t o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
  int64_t z2 = o0 + o2, z3 = o1 + o3;
  z1 = (z2 + z3) * F_1_175875602;
  z2 = z2 * -F_1_961570560 + z1;
  z3 = z3 * -F_0_390180644 + z1;
  z1 = (o0 + o3) * -F_0_899976223;
  o0 = o0 * F_0_298631336 + z1 + z2;
  o3 = o3 * F_1_501321110 + z1 + z3;
  z1 = (o1 + o2) * -F_2_562915447;
  o1 = o1 * F_2_053119869 + z1 + z3;
  o2 = o2 * F_3_072711026 + z1 + z2;
  out[0] = (int32_t)((t10 + o3) >> shift); out[7] = (int32_t)((t10 - o3) >> shift);
  out[1] = (int32_t)((t11 + o2) >> shift); out[6] = (int32_t)((t11 - o2) >> shift);
  out[2] = (int32_t)((t12 + o1) >> shift); out[5] = (int32_t)((t12 - o1) >> shift);
  out[3] = (int32_t)((t13 + o0) >> shift); out[4] = (int32_t)((t13 - o0) >> shift);
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
    idct1d(ws + 8 * y, res, ((int64_t)128 << (CB + P1 + 3)) + (1 << (CB + P1 + 2)), CB + P1 + 3);
    for (int x = 0; x < 8; ++x) dst[y * stride + x] = clamp8(res[x]);
  }
}

/* ---- 16 output samples from the 8 coefficients of a block: the larger inverse transform libjpeg (7 and later)
 * brings subsampled chroma to full resolution with, in the same fixed point as the 8-point one. cK = sqrt(2)
 * cos(K pi / 32); the constants are those combinations rounded to 13 bits. */
#define FX(x) ((int64_t)((x) * (1 << CB) + 0.5))
static void idct1d_16(const int32_t in[8], int32_t out[16], int64_t round, int shift) {
  int64_t t0 = (int64_t)in[0] * (1 << CB) + round;
  int64_t z1 = in[4];
  int64_t t1 = z1 * FX(1.306562965), t2 = z1 * F_0_541196100;
  const int64_t t10 = t0 + t1, t11 = t0 - t1, t12 = t0 + t2, t13 = t0 - t2;
  z1 = in[2];
  int64_t z2 = in[6], z3 = z1 - z2;
  int64_t z4 = z3 * FX(0.275899379);
  z3 = z3 * FX(1.387039845);
  t0 = z3 + z2 * F_2_562915447;
  t1 = z4 + z1 * F_0_899976223;
  t2 = z3 - z1 * FX(0.601344887);
  int64_t t3 = z4 - z2 * FX(0.509795579);
  const int64_t t20 = t10 + t0, t27 = t10 - t0, t21 = t12 + t1, t26 = t12 - t1;
  const int64_t t22 = t13 + t2, t25 = t13 - t2, t23 = t11 + t3, t24 = t11 - t3;
  z1 = in[1]; z2 = in[3]; z3 = in[5]; z4 = in[7];
  int64_t o11 = z1 + z3;
  int64_t o1 = (z1 + z2) * FX(1.353318001), o2 = o11 * FX(1.247225013), o3 = (z1 + z4) * FX(1.093201867);
  int64_t o10 = (z1 - z4) * FX(0.897167586);
  o11 = o11 * FX(0.666655658);
  int64_t o12 = (z1 - z2) * FX(0.410524528);
  const int64_t o0 = o1 + o2 + o3 - z1 * FX(2.286341144);
  const int64_t o13 = o10 + o11 + o12 - z1 * FX(1.835730603);
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
  out[0] = (int32_t)((t20 + o0) >> shift);   out[15] = (int32_t)((t20 - o0) >> shift);
  out[1] = (int32_t)((t21 + o1) >> shift);   out[14] = (int32_t)((t21 - o1) >> shift);
  out[2] = (int32_t)((t22 + o2) >> shift);   out[13] = (int32_t)((t22 - o2) >> shift);
  out[3] = (int32_t)((t23 + o3) >> shift);   out[12] = (int32_t)((t23 - o3) >> shift);
  out[4] = (int32_t)((t24 + o10) >> shift);  out[11] = (int32_t)((t24 - o10) >> shift);
  out[5] = (int32_t)((t25 + o11) >> shift);  out[10] = (int32_t)((t25 - o11) >> shift);
  out[6] = (int32_t)((t26 + o12) >> shift);  out[9] = (int32_t)((t26 - o12) >> shift);
  out[7] = (int32_t)((t27 + o13) >> shift);  out[8] = (int32_t)((t27 - o13) >> shift);
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
    if (fx == 2) idct1d_16(ws + 8 * y, res, ((int64_t)128 << (CB + P1 + 3)) + (1 << (CB + P1 + 2)), CB + P1 + 3);
    else idct1d(ws + 8 * y, res, ((int64_t)128 << (CB + P1 + 3)) + (1 << (CB + P1 + 2)), CB + P1 + 3);
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
  int id, hs, vs, tq;
  int bw, bh;        /* blocks per row / column, padded to whole MCUs */
  int nbw, nbh;      /* blocks that really cover the component: the geometry of a scan of its own */
  int32_t *coef;     /* [bh][bw][64] in natural order, as decoded (not yet multiplied by the quantisation table) */
  uint32_t q[64];    /* its quantisation table, latched when its first scan starts */
  int have_q;
  int pred;
  unsigned char *pix; /* full resolution, padded */
  size_t stride;
};

struct scan {
  int ns, ci[3], td[3], ta[3], ss, se, ah, al;
};

/* one block of one scan; returns 0, or -1 on corrupt data. `eobrun` = blocks still covered by an end-of-band run */
static int scan_block(struct bits *st, const struct scan *sc, int k0, struct comp *c, int32_t *blk, const struct huff *dc,
                      const struct huff *ac, int progressive, int *eobrun) {
  if (!progressive) { /* sequential: the whole block */
    int t = huff_decode(st, dc);
    if (t < 0 || t > 11) return -1;
    c->pred = (int)((unsigned)c->pred + (unsigned)extend(bits_get(st, t), t)); /* (hostile data: wraps, never overflows) */
    blk[0] = c->pred;
    for (int k = 1; k < 64;) {
      const int rs = huff_decode(st, ac);
      if (rs < 0) return -1;
      const int r = rs >> 4, sz = rs & 15;
      if (!sz) {
        if (r == 15) { k += 16; continue; }
        break;
      }
      k += r;
      if (k > 63) return -1;
      blk[ZIGZAG[k]] = extend(bits_get(st, sz), sz);
      ++k;
    }
    return 0;
  }
  (void)k0;
  if (sc->ss == 0) { /* DC: first pass (the value, shifted) or one more bit */
    if (sc->ah == 0) {
      int t = huff_decode(st, dc);
      if (t < 0 || t > 11) return -1;
      c->pred = (int)((unsigned)c->pred + (unsigned)extend(bits_get(st, t), t));
      blk[0] = (int32_t)((uint32_t)c->pred << sc->al);
    } else if (bits_get(st, 1)) {
      blk[0] |= 1 << sc->al;
    }
    return 0;
  }
  if (sc->ah == 0) { /* AC, first pass over the band ss..se */
    if (*eobrun > 0) { --*eobrun; return 0; }
    for (int k = sc->ss; k <= sc->se;) {
      const int rs = huff_decode(st, ac);
      if (rs < 0) return -1;
      const int r = rs >> 4, sz = rs & 15;
      if (!sz) {
        if (r < 15) { *eobrun = (1 << r) - 1 + (r ? bits_get(st, r) : 0); break; }
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return -1;
      blk[ZIGZAG[k]] = (int32_t)((uint32_t)extend(bits_get(st, sz), sz) << sc->al);
      ++k;
    }
    return 0;
  }
  /* AC, refinement: one more bit for the coefficients that are known, new ones of size one in between */
  const int32_t p1 = 1 << sc->al, m1 = -(1 << sc->al);
  int k = sc->ss;
  if (*eobrun == 0) {
    for (; k <= sc->se; ++k) {
      const int rs = huff_decode(st, ac);
      if (rs < 0) return -1;
      int r = rs >> 4;
      int32_t val = 0;
      if (rs & 15) {
        if ((rs & 15) != 1) return -1;
        val = bits_get(st, 1) ? p1 : m1;
      } else if (r != 15) {
        *eobrun = (1 << r) + (r ? bits_get(st, r) : 0);
        break;
      }
      for (; k <= sc->se; ++k) { /* pass r zero-history coefficients; known ones take a correction bit on the way */
        int32_t *cp = &blk[ZIGZAG[k]];
        if (*cp) {
          if (bits_get(st, 1) && !(*cp & p1)) *cp += *cp >= 0 ? p1 : m1;
        } else if (--r < 0) {
          break;
        }
      }
      if (val && k <= sc->se) blk[ZIGZAG[k]] = val;
    }
  }
  if (*eobrun > 0) {
    for (; k <= sc->se; ++k) {
      int32_t *cp = &blk[ZIGZAG[k]];
      if (*cp && bits_get(st, 1) && !(*cp & p1)) *cp += *cp >= 0 ? p1 : m1;
    }
    --*eobrun;
  }
  return 0;
}

float *nlk_read_jpeg(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  uint32_t qt[4][64];
  int have_qt[4] = {0, 0, 0, 0};
  struct huff dc[4], ac[4];
  struct comp C[3];
  memset(dc, 0, sizeof dc);
  memset(ac, 0, sizeof ac);
  memset(C, 0, sizeof C);
  int nc = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, have_sof = 0, progressive = 0, plain_rgb = 0, nscans = 0;
  int mx = 0, my = 0;
  const char *err = NULL;
  float *out = NULL;
  size_t i = 2;
  while (!err && i + 4 <= n) {
    if (b[i] != 0xFF) { ++i; continue; }
    const int m = b[i + 1];
    if (m == 0xFF) { ++i; continue; }
    if (m == 0xD8 || m == 0x01 || m == 0x00 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
    if (m == 0xD9) break;
    const size_t len = (size_t)b[i + 2] << 8 | b[i + 3];
    if (len < 2 || i + 2 + len > n) { err = "truncated JPEG segment"; break; }
    const unsigned char *s = b + i + 4, *e = b + i + 2 + len;
    i += 2 + len;
    if (m == 0xDB) { /* quantisation tables */
      while (s < e && !err) {
        const int pq = s[0] >> 4, tq = s[0] & 15;
        if (tq > 3 || pq > 1 || s + 1 + 64 * (pq + 1) > e) { err = "bad JPEG quantisation table"; break; }
        ++s;
        for (int k = 0; k < 64; ++k, s += pq + 1) qt[tq][ZIGZAG[k]] = pq ? (uint32_t)(s[0] << 8 | s[1]) : s[0];
        have_qt[tq] = 1;
      }
    } else if (m == 0xC4) { /* Huffman tables */
      while (s < e && !err) {
        const int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3 || s + 17 > e) { err = "bad JPEG Huffman table"; break; }
        struct huff *t = tc ? &ac[th] : &dc[th];
        int total = 0;
        t->counts[0] = 0;
        for (int l = 1; l <= 16; ++l) total += (t->counts[l] = s[l]);
        if (total > 256 || s + 17 + total > e) { err = "bad JPEG Huffman table"; break; }
        memcpy(t->symbols, s + 17, (size_t)total);
        huff_build(t);
        t->present = 1;
        s += 17 + total;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { /* frame header: sequential or progressive DCT, Huffman */
      if (have_sof || e - s < 6) { err = "bad JPEG frame header"; break; }
      if (s[0] != 8) { err = "JPEG precision other than 8 bits is not supported"; break; }
      progressive = m == 0xC2;
      H = s[1] << 8 | s[2];
      W = s[3] << 8 | s[4];
      nc = s[5];
      if ((nc != 1 && nc != 3) || e - s < 6 + 3 * nc) { err = "JPEG with other than 1 or 3 components is not supported"; break; }
      if (W <= 0 || H <= 0 || W > 65500 || H > 65500) { err = "bad JPEG size"; break; }
      for (int c = 0; c < nc; ++c) {
        C[c].id = s[6 + 3 * c];
        C[c].hs = s[7 + 3 * c] >> 4;
        C[c].vs = s[7 + 3 * c] & 15;
        C[c].tq = s[8 + 3 * c];
        if (C[c].hs < 1 || C[c].hs > 4 || C[c].vs < 1 || C[c].vs > 4 || C[c].tq > 3) err = "bad JPEG sampling factors";
        if (C[c].hs > hmax) hmax = C[c].hs;
        if (C[c].vs > vmax) vmax = C[c].vs;
      }
      if (err) break;
      if (nc == 1) C[0].hs = C[0].vs = hmax = vmax = 1; /* (a single component is never interleaved) */
      for (int c = 0; c < nc; ++c)
        if (hmax % C[c].hs || vmax % C[c].vs || hmax / C[c].hs > 4 || vmax / C[c].vs > 4)
          err = "JPEG sampling factors that do not divide each other are not supported";
      if (err) break;
      /* the same bound as every other format (imgio.c: sane_size), and the file must be able to hold the image at
         all: a 200-byte header that announces 65500 x 65500 must not make this process allocate 17 GB (inside
         nlk-server that would take the resident process down). Every 8 x 8 block of every component costs at least
         one bit - its DC code, baseline or progressive (a flat progressive file with optimised tables spends little
         more: its AC scans are end-of-band runs) - so the blocks are counted from the sampling factors (a 4:2:0 file
         has half the blocks of a 4:4:4 one: ADVICE r5) and the file must hold HALF a bit for each. */
      uint64_t nblocks = 0;
      for (int c = 0; c < nc; ++c)
        nblocks += (uint64_t)((((uint64_t)W * C[c].hs + hmax - 1) / hmax + 7) / 8) *
                   (uint64_t)((((uint64_t)H * C[c].vs + vmax - 1) / vmax + 7) / 8);
      if ((uint64_t)W * (uint64_t)H * (uint64_t)nc > ((uint64_t)1 << 31) || nblocks / 16 > (uint64_t)n + 1024) {
        err = "JPEG size is unreasonable for the size of the file";
        break;
      }
      const int mcuw = 8 * hmax, mcuh = 8 * vmax;
      mx = (W + mcuw - 1) / mcuw;
      my = (H + mcuh - 1) / mcuh;
      for (int c = 0; c < nc; ++c) {
        C[c].bw = mx * C[c].hs;
        C[c].bh = my * C[c].vs;
        C[c].nbw = ((W * C[c].hs + hmax - 1) / hmax + 7) / 8;
        C[c].nbh = ((H * C[c].vs + vmax - 1) / vmax + 7) / 8;
        C[c].coef = calloc((size_t)C[c].bw * C[c].bh * 64, sizeof(int32_t));
        C[c].stride = (size_t)mx * mcuw;
        C[c].pix = malloc(C[c].stride * (size_t)my * mcuh);
        if (!C[c].coef || !C[c].pix) err = "out of memory";
      }
      have_sof = 1;
    } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      err = "lossless / hierarchical / arithmetic-coded JPEG is not supported";
    } else if (m == 0xEE) { /* Adobe: transform 0 = the three components ARE red, green, blue */
      if (e - s >= 12 && !memcmp(s, "Adobe", 5)) plain_rgb = s[11] == 0;
    } else if (m == 0xDD) {
      if (e - s < 2) { err = "bad JPEG restart interval"; break; }
      restart = s[0] << 8 | s[1];
    } else if (m == 0xDA) { /* a scan */
      struct scan sc;
      memset(&sc, 0, sizeof sc);
      if (!have_sof || e - s < 1) { err = "JPEG scan before the frame header"; break; }
      sc.ns = s[0];
      if (sc.ns < 1 || sc.ns > nc || e - s < 1 + 2 * sc.ns + 3) { err = "bad JPEG scan header"; break; }
      for (int k = 0; k < sc.ns && !err; ++k) {
        int c = 0;
        while (c < nc && C[c].id != s[1 + 2 * k]) ++c;
        if (c == nc) { err = "bad JPEG scan header"; break; }
        sc.ci[k] = c;
        sc.td[k] = s[2 + 2 * k] >> 4;
        sc.ta[k] = s[2 + 2 * k] & 15;
        if (sc.td[k] > 3 || sc.ta[k] > 3) err = "bad JPEG scan header";
        if (!C[c].have_q) { /* the table in force when the component is first scanned is the component's */
          if (!have_qt[C[c].tq]) { err = "JPEG scan uses a quantisation table the file does not define"; break; }
          memcpy(C[c].q, qt[C[c].tq], sizeof C[c].q);
          C[c].have_q = 1;
        }
      }
      if (err) break;
      sc.ss = s[1 + 2 * sc.ns];
      sc.se = s[2 + 2 * sc.ns];
      sc.ah = s[3 + 2 * sc.ns] >> 4;
      sc.al = s[3 + 2 * sc.ns] & 15;
      if (!progressive) { sc.ss = 0; sc.se = 63; sc.ah = sc.al = 0; }
      if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || sc.ah > 13 || (sc.ss == 0 && sc.se != 0 && progressive) ||
          (sc.ss > 0 && sc.ns != 1)) { err = "bad JPEG scan parameters"; break; }
      for (int k = 0; k < sc.ns; ++k) {
        const int need_dc = !progressive || sc.ss == 0, need_ac = !progressive || sc.ss > 0;
        if ((need_dc && !(progressive && sc.ah) && !dc[sc.td[k]].present) || (need_ac && !ac[sc.ta[k]].present))
          err = "JPEG scan uses a Huffman table the file does not define";
        C[sc.ci[k]].pred = 0;
      }
      if (err) break;
      struct bits st = {b + i, b + n, 0, 0, 0};
      int until_restart = restart, eobrun = 0, bad = 0;
      /* a scan of one component walks that component's own blocks; several components are interleaved in MCUs */
      const int single = sc.ns == 1;
      struct comp *c0 = &C[sc.ci[0]];
      const int units_x = single ? c0->nbw : mx, units = single ? c0->nbw * c0->nbh : mx * my;
      for (int u = 0; u < units && !bad; ++u) {
        if (restart && until_restart == 0) { /* byte-align, pass the RSTn marker, reset predictions and runs */
          st.cnt = 0;
          st.acc = 0;
          if (!(st.marker >= 0xD0 && st.marker <= 0xD7))
            while (st.p + 1 < st.end && !(st.p[0] == 0xFF && st.p[1] >= 0xD0 && st.p[1] <= 0xD7)) ++st.p;
          st.p += 2;
          if (st.p > st.end) st.p = st.end;
          st.marker = 0;
          for (int k = 0; k < sc.ns; ++k) C[sc.ci[k]].pred = 0;
          eobrun = 0;
          until_restart = restart;
        }
        --until_restart;
        const int ux = u % units_x, uy = u / units_x;
        if (single) {
          int32_t *blk = c0->coef + ((size_t)uy * c0->bw + ux) * 64;
          bad = scan_block(&st, &sc, 0, c0, blk, &dc[sc.td[0]], &ac[sc.ta[0]], progressive, &eobrun);
        } else {
          for (int k = 0; k < sc.ns && !bad; ++k) {
            struct comp *c = &C[sc.ci[k]];
            for (int v = 0; v < c->vs && !bad; ++v)
              for (int hh = 0; hh < c->hs && !bad; ++hh) {
                int32_t *blk = c->coef + ((size_t)(uy * c->vs + v) * c->bw + (ux * c->hs + hh)) * 64;
                bad = scan_block(&st, &sc, 0, c, blk, &dc[sc.td[k]], &ac[sc.ta[k]], progressive, &eobrun);
              }
          }
        }
      }
      if (bad) { err = "corrupt JPEG data"; break; }
      ++nscans;
      /* on to the marker that ends the entropy-coded data */
      const unsigned char *p = st.p;
      while (p + 1 < b + n && !(p[0] == 0xFF && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7))) ++p;
      i = (size_t)(p - b);
    }
  }
  if (!err && !nscans) err = "JPEG without a scan";
  if (!err) { /* samples: every component at full resolution */
    for (int c = 0; c < nc; ++c) {
      const int fx = hmax / C[c].hs, fy = vmax / C[c].vs;
      int32_t deq[64];
      for (int by = 0; by < C[c].bh; ++by)
        for (int bx = 0; bx < C[c].bw; ++bx) {
          const int32_t *blk = C[c].coef + ((size_t)by * C[c].bw + bx) * 64;
          for (int k = 0; k < 64; ++k) deq[k] = (int32_t)((uint32_t)blk[k] * C[c].q[k]);
          unsigned char *dst = C[c].pix + (size_t)by * 8 * fy * C[c].stride + (size_t)bx * 8 * fx;
          if (fx == 1 && fy == 1) idct8x8(deq, dst, C[c].stride);
          else block_to_full(deq, fx, fy, dst, C[c].stride);
        }
    }
    out = malloc((size_t)W * H * nc * sizeof(float));
    if (!out) err = "out of memory";
  }
  if (!err) {
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
  for (int c = 0; c < 3; ++c) { free(C[c].coef); free(C[c].pix); }
  if (err) return jfail(path, err);
  *w = W; *h = H; *ch = nc;
  return out;
}
