/* imgio.c — see imgio.h. Own code: TIFF/PNG/PFM/FLO codecs written from the
 * public format specifications (TIFF 6.0 + BigTIFF, PNG 1.2, PFM, Middlebury
 * .flo); zlib's uncompress() is reached through dlopen for Deflate/PNG. */
#include "imgio.h"

#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

float *nlk_read_jpeg(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch); /* imgio_jpeg.c */

static float *fail(const char *path, const char *why) {
  fprintf(stderr, "imgio: %s: %s\n", path, why);
  return NULL;
}

/* header-supplied sizes: positive and at most 2^31 samples (8 GiB of floats) */
static int sane_size(uint64_t w, uint64_t h, uint64_t ch) {
  return w > 0 && h > 0 && ch > 0 && w <= 1u << 20 && h <= 1u << 20 && ch <= 1024 && w * h <= (1ull << 31) / ch;
}

static unsigned char *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < 0) { fclose(f); return NULL; }
  unsigned char *b = malloc((size_t)sz + 1); /* + a NUL: the text headers are parsed with sscanf */
  if (b && fread(b, 1, (size_t)sz, f) != (size_t)sz) { free(b); b = NULL; }
  if (b) b[sz] = 0;
  fclose(f);
  *n = (size_t)sz;
  return b;
}

/* ------------------------------------------------------------------- zlib */
typedef int (*uncompress_fn)(unsigned char *, unsigned long *, const unsigned char *, unsigned long);
static int z_uncompress(unsigned char *dst, size_t *dlen, const unsigned char *src, size_t slen) {
  static uncompress_fn fn = NULL;
  if (!fn) {
    void *h = dlopen("libz.so.1", RTLD_NOW);
    if (!h) h = dlopen("libz.so", RTLD_NOW);
    if (h) fn = (uncompress_fn)dlsym(h, "uncompress");
    if (!fn) { fprintf(stderr, "imgio: zlib (libz.so.1) not found\n"); return -1; }
  }
  unsigned long n = *dlen;
  int rc = fn(dst, &n, src, slen);
  *dlen = n;
  return rc == 0 || rc == -5 /* Z_BUF_ERROR with full output is fine for us */ ? 0 : -1;
}

/* -------------------------------------------------------------------- PFM */
static float *read_pfm(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  int c = b[1] == 'F' ? 3 : 1, pos = 0;
  double scale;
  if (sscanf((const char *)b + 2, "%d %d %lf%n", w, h, &scale, &pos) != 3) return fail(path, "bad PFM header");
  size_t off = 2 + pos + 1; /* one whitespace byte after the scale */
  if (*w <= 0 || *h <= 0 || !sane_size((uint64_t)*w, (uint64_t)*h, (uint64_t)c)) return fail(path, "bad PFM size");
  size_t cnt = (size_t)*w * *h * c;
  if (off + cnt * 4 > n) return fail(path, "truncated PFM");
  float *d = malloc(cnt * 4);
  if (!d) return fail(path, "out of memory");
  memcpy(d, b + off, cnt * 4);
  if (scale > 0) /* big-endian samples */
    for (size_t i = 0; i < cnt; ++i) {
      unsigned char *p = (unsigned char *)&d[i], t;
      t = p[0]; p[0] = p[3]; p[3] = t; t = p[1]; p[1] = p[2]; p[2] = t;
    }
  *ch = c;
  return d; /* top row first, no flip (reference: lib/iio/iio.c:2049-2071) */
}

/* -------------------------------------------------------------------- PNM */
/* P2 / P3 (ASCII) and P5 / P6 (binary; two big-endian bytes per sample when maxval > 255) grey and colour maps,
 * read the way the reference's library reads them (lib/iio/iio.c:1759-1805: comments between the header fields,
 * one white-space byte after maxval, maxval itself not applied - samples stay what the file says; ASCII samples
 * may be any number strtof takes). */
static size_t pnm_skip(const unsigned char *b, size_t n, size_t i) { /* white space and # comments */
  while (i < n) {
    if (b[i] == '#') { while (i < n && b[i] != '\n') ++i; }
    else if (b[i] == ' ' || (b[i] >= '\t' && b[i] <= '\r')) ++i;
    else break;
  }
  return i;
}
static int pnm_int(const unsigned char *b, size_t n, size_t *i, long *v) {
  *i = pnm_skip(b, n, *i);
  if (*i >= n || b[*i] < '0' || b[*i] > '9') return -1;
  long x = 0;
  while (*i < n && b[*i] >= '0' && b[*i] <= '9' && x < (1L << 40)) x = x * 10 + (b[(*i)++] - '0');
  *v = x;
  return 0;
}
static float *read_pnm(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  const int kind = b[1] - '0', c = (kind == 3 || kind == 6) ? 3 : 1, ascii = kind == 2 || kind == 3;
  size_t i = 2;
  long ww, hh, maxval;
  if (pnm_int(b, n, &i, &ww) || pnm_int(b, n, &i, &hh) || pnm_int(b, n, &i, &maxval)) return fail(path, "bad PNM header");
  if (i >= n || !(b[i] == ' ' || (b[i] >= '\t' && b[i] <= '\r'))) return fail(path, "bad PNM header");
  ++i; /* exactly one white-space byte, then the samples */
  if (ww <= 0 || hh <= 0 || ww > INT32_MAX || hh > INT32_MAX || !sane_size((uint64_t)ww, (uint64_t)hh, (uint64_t)c))
    return fail(path, "bad PNM size");
  if (maxval <= 0 || maxval >= 65536) return fail(path, "PNM maxval out of range");
  const size_t cnt = (size_t)ww * hh * c;
  const int wide = maxval > 255;
  if (!ascii && i + cnt * (wide ? 2 : 1) > n) return fail(path, "truncated PNM");
  if (ascii && cnt > n) return fail(path, "truncated PNM"); /* (a sample takes at least one byte) */
  float *d = malloc(cnt * 4);
  if (!d) return fail(path, "out of memory");
  if (!ascii) {
    const unsigned char *q = b + i;
    for (size_t k = 0; k < cnt; ++k) d[k] = wide ? (float)(q[2 * k] * 256 + q[2 * k + 1]) : (float)q[k];
  } else {
    /* (the buffer is NUL-terminated by slurp(): strtof stops there at the latest) */
    const char *q = (const char *)b + i;
    for (size_t k = 0; k < cnt; ++k) {
      char *end;
      d[k] = strtof(q, &end);
      if (end == q) { free(d); return fail(path, "truncated PNM"); }
      q = end;
    }
  }
  *w = (int)ww; *h = (int)hh; *ch = c;
  return d;
}

/* -------------------------------------------------------------------- FLO */
static float *read_flo(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  int32_t ww, hh;
  memcpy(&ww, b + 4, 4);
  memcpy(&hh, b + 8, 4);
  if (ww <= 0 || hh <= 0 || !sane_size((uint64_t)ww, (uint64_t)hh, 2)) return fail(path, "bad .flo");
  size_t cnt = (size_t)ww * hh * 2;
  if (12 + cnt * 4 > n) return fail(path, "bad .flo");
  float *d = malloc(cnt * 4);
  if (!d) return fail(path, "out of memory");
  memcpy(d, b + 12, cnt * 4);
  *w = ww; *h = hh; *ch = 2;
  return d;
}

/* ------------------------------------------------------------------- TIFF */
typedef struct { const unsigned char *b; size_t n; int le, big; } tif_t;
static uint64_t rd(const tif_t *t, size_t off, int bytes) {
  uint64_t v = 0;
  if (off + bytes > t->n) return 0;
  for (int i = 0; i < bytes; ++i)
    v |= (uint64_t)t->b[off + (t->le ? i : bytes - 1 - i)] << (8 * i);
  return v;
}
static const int tif_tsz[] = {0, 1, 1, 2, 4, 8, 1, 1, 2, 4, 8, 4, 8, 4, 0, 0, 8, 8, 8};

/* value j of an IFD entry located at `e` */
static uint64_t tif_val(const tif_t *t, size_t e, uint64_t j) {
  const int type = (int)rd(t, e + 2, 2);
  const uint64_t cnt = rd(t, e + 4, t->big ? 8 : 4);
  const int ts = type < 19 ? tif_tsz[type] : 0;
  const size_t field = e + (t->big ? 12 : 8), fsz = t->big ? 8 : 4;
  const size_t base = (cnt * ts <= fsz) ? field : (size_t)rd(t, field, (int)fsz);
  return j < cnt ? rd(t, base + j * ts, ts) : 0;
}

/* TIFF LZW: MSB-first codes of 9..12 bits, early change, 256 = clear, 257 = end of data.
 * Returns the number of bytes decoded, or (size_t)-1 on a corrupt stream (a code beyond the
 * next free entry, or a code before any clear-state literal). Every table entry that can be
 * reached was written since the last clear code, and every prefix is smaller than its entry,
 * so the chain walks terminate within the 4096-byte stack. */
static size_t lzw_decode(const unsigned char *s, size_t n, unsigned char *d, size_t cap) {
  uint16_t prefix[4096];
  unsigned char suffix[4096], stack[4096];
  size_t out = 0, bitpos = 0;
  int width = 9, next = 258, prev = -1;
  for (;;) {
    if (bitpos + width > 8 * n) break; /* ran out of bits without an end code: tolerated (got < want is caught by the caller) */
    uint32_t code = 0;
    for (int i = 0; i < width; ++i) {
      const size_t bp = bitpos + i;
      code = (code << 1) | ((s[bp / 8] >> (7 - bp % 8)) & 1);
    }
    bitpos += width;
    if (code == 257) break;
    if (code == 256) { width = 9; next = 258; prev = -1; continue; }
    if (prev < 0) { /* first code after a clear must be a literal */
      if (code > 255) return (size_t)-1;
      if (out < cap) d[out++] = (unsigned char)code;
      prev = (int)code;
      continue;
    }
    if ((int)code > next || (int)code >= 4096) return (size_t)-1; /* only code == next (KwKwK) may be undefined */
    int sp = 0, c = (int)code;
    if (c == next) { /* KwKwK: the string of prev followed by its own first byte */
      int p = prev;
      while (p >= 258 && sp < 4095) p = prefix[p];
      if (p >= 258) return (size_t)-1;
      stack[sp++] = (unsigned char)p;
      c = prev;
    }
    while (c >= 258 && sp < 4095) { stack[sp++] = suffix[c]; c = prefix[c]; }
    if (c >= 258) return (size_t)-1;
    stack[sp++] = (unsigned char)c;
    const unsigned char first = (unsigned char)c;
    while (sp > 0 && out < cap) d[out++] = stack[--sp];
    if (next < 4096) {
      prefix[next] = (uint16_t)prev; /* prev < next: chains strictly decrease */
      suffix[next] = first;
      next++;
      if (next == 511 || next == 1023 || next == 2047) width++;
    }
    prev = (int)code;
    if (out >= cap) break;
  }
  return out;
}

/* the same code stream the other way (for the writer): greedy longest match, hash-probed table */
static size_t lzw_encode(const unsigned char *s, size_t n, unsigned char *d, size_t cap) {
  enum { HS = 9001 };
  static __thread int32_t hkey[HS];
  static __thread uint16_t hval[HS];
  size_t o = 0;
  uint64_t acc = 0;
  int nbits = 0, width = 9, next = 258;
#define LZW_PUT(code) do { acc = (acc << width) | (uint32_t)(code); nbits += width; \
    while (nbits >= 8) { if (o < cap) d[o] = (unsigned char)(acc >> (nbits - 8)); ++o; nbits -= 8; } } while (0)
  for (int i = 0; i < HS; ++i) hkey[i] = -1;
  LZW_PUT(256);
  if (n == 0) { LZW_PUT(257); if (nbits) { if (o < cap) d[o] = (unsigned char)(acc << (8 - nbits)); ++o; } return o; }
  int cur = s[0];
  for (size_t i = 1; i < n; ++i) {
    const int ch = s[i];
    const int32_t key = (cur << 8) | ch;
    int h = (int)(((uint32_t)key * 2654435761u) % HS);
    while (hkey[h] != -1 && hkey[h] != key) h = h + 1 == HS ? 0 : h + 1;
    if (hkey[h] == key) { cur = hval[h]; continue; }
    LZW_PUT(cur);
    hkey[h] = key; hval[h] = (uint16_t)next++;
    /* the decoder creates each entry one code later than the encoder, so its "early" widening at
     * 511 / 1023 / 2047 entries is the encoder's at 512 / 1024 / 2048 */
    if (next == 4094) { /* table full: clear */
      LZW_PUT(256);
      for (int k = 0; k < HS; ++k) hkey[k] = -1;
      width = 9; next = 258;
    } else if (next == 512 || next == 1024 || next == 2048) width++;
    cur = ch;
  }
  LZW_PUT(cur);
  next++; /* the decoder adds an entry for this code too, and may widen (or fill up) because of it */
  if (next == 4094) { LZW_PUT(256); width = 9; }
  else if (next == 512 || next == 1024 || next == 2048) width++;
  LZW_PUT(257);
  if (nbits) { if (o < cap) d[o] = (unsigned char)(acc << (8 - nbits)); ++o; }
#undef LZW_PUT
  return o;
}

static size_t packbits_decode(const unsigned char *s, size_t n, unsigned char *d, size_t cap) {
  size_t i = 0, o = 0;
  while (i < n && o < cap) {
    int c = (signed char)s[i++];
    if (c >= 0) { for (int k = 0; k <= c && i < n && o < cap; ++k) d[o++] = s[i++]; }
    else if (c != -128) { for (int k = 0; k <= -c && o < cap; ++k) d[o++] = s[i]; i++; }
  }
  return o;
}

static float *read_tiff(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  tif_t t = {b, n, b[0] == 'I', 0};
  const int magic = (int)rd(&t, 2, 2);
  if (magic == 43) t.big = 1;
  else if (magic != 42) return fail(path, "not a TIFF");
  size_t ifd = (size_t)(t.big ? rd(&t, 8, 8) : rd(&t, 4, 4));
  const uint64_t nent = t.big ? rd(&t, ifd, 8) : rd(&t, ifd, 2);
  const size_t e0 = ifd + (t.big ? 8 : 2), esz = t.big ? 20 : 12;
  if (ifd >= n || nent > 4096 || e0 + nent * esz > n) return fail(path, "TIFF directory outside the file");
  uint64_t W = 0, H = 0, bps = 1, comp = 1, spp = 1, rps = 0, planar = 1, fmt = 1, pred = 1, TW = 0, TL = 0;
  size_t e_off = 0, e_cnt = 0, e_toff = 0, e_tcnt = 0;
  for (uint64_t i = 0; i < nent; ++i) {
    const size_t e = e0 + i * esz;
    switch ((int)rd(&t, e, 2)) {
      case 256: W = tif_val(&t, e, 0); break;
      case 257: H = tif_val(&t, e, 0); break;
      case 258: bps = tif_val(&t, e, 0); break;
      case 259: comp = tif_val(&t, e, 0); break;
      case 273: e_off = e; break;
      case 277: spp = tif_val(&t, e, 0); break;
      case 278: rps = tif_val(&t, e, 0); break;
      case 279: e_cnt = e; break;
      case 284: planar = tif_val(&t, e, 0); break;
      case 317: pred = tif_val(&t, e, 0); break;
      case 339: fmt = tif_val(&t, e, 0); break;
      case 322: TW = tif_val(&t, e, 0); break;
      case 323: TL = tif_val(&t, e, 0); break;
      case 324: e_toff = e; break;
      case 325: e_tcnt = e; break;
    }
  }
  /* Data is organised either in strips of `rps` rows or in TW x TL tiles (TIFF 6.0 section 15;
   * the reference's reader takes both through libtiff, lib/iio/iio.c:1463-1661). Both are handled as
   * "chunks" of cw x cl pixels, `across` of them per row of chunks; edge tiles are stored whole. */
  const int tiled = TW && TL && e_toff && e_tcnt;
  if (tiled) { e_off = e_toff; e_cnt = e_tcnt; }
  if (!W || !H || !e_off || !e_cnt) return fail(path, "incomplete TIFF directory");
  if (!sane_size(W, H, spp)) return fail(path, "unreasonable TIFF size");
  if (!rps || rps > H) rps = H;
  if (bps != 8 && bps != 16 && bps != 32 && bps != 64) return fail(path, "unsupported bits per sample");
  if (pred != 1 && pred != 2 && pred != 3) return fail(path, "unknown TIFF predictor");
  if (pred == 3 && (fmt != 3 || (bps != 32 && bps != 64))) return fail(path, "floating-point predictor on non-float samples");
  if (tiled && (TW > (1u << 20) || TL > (1u << 20))) return fail(path, "unreasonable tile size");
  const int bytes = (int)bps / 8;
  const uint64_t planes = planar == 2 ? spp : 1, cpp = planar == 2 ? 1 : spp; /* comps per pixel in a chunk */
  const uint64_t cw = tiled ? TW : W, cl = tiled ? TL : rps;
  const uint64_t across = (W + cw - 1) / cw, down = (H + cl - 1) / cl;
  float *out = malloc((size_t)W * H * spp * sizeof(float));
  unsigned char *raw = malloc((size_t)cw * cl * cpp * bytes + 16);
  if (!out || !raw) { free(out); free(raw); return fail(path, "out of memory"); }
  for (uint64_t pl = 0; pl < planes; ++pl)
    for (uint64_t cy = 0; cy < down; ++cy)
      for (uint64_t cx = 0; cx < across; ++cx) {
        const uint64_t idx = (pl * down + cy) * across + cx;
        const size_t off = (size_t)tif_val(&t, e_off, idx), cnt = (size_t)tif_val(&t, e_cnt, idx);
        /* a strip holds only the rows that exist, a tile is always whole */
        const uint64_t rows_in = tiled ? cl : ((cy + 1) * cl <= H ? cl : H - cy * cl);
        const uint64_t rows = (cy + 1) * cl <= H ? cl : H - cy * cl;   /* rows of it inside the image */
        const uint64_t cols = (cx + 1) * cw <= W ? cw : W - cx * cw;
        const size_t want = (size_t)cw * rows_in * cpp * bytes;
        if (off > n || cnt > n - off) { free(out); free(raw); return fail(path, "strip / tile outside the file"); }
        size_t got = want;
        if (comp == 1) { got = cnt < want ? cnt : want; memcpy(raw, b + off, got); }
        else if (comp == 5) got = lzw_decode(b + off, cnt, raw, want);
        else if (comp == 32773) got = packbits_decode(b + off, cnt, raw, want);
        else if (comp == 8 || comp == 32946) { if (z_uncompress(raw, &got, b + off, cnt)) got = 0; }
        else { free(out); free(raw); return fail(path, "unsupported TIFF compression"); }
        if (got == (size_t)-1) { free(out); free(raw); return fail(path, "corrupt LZW stream"); }
        if (got < want) { free(out); free(raw); return fail(path, "short strip / tile"); }
        if (pred == 3) {
          /* floating-point predictor (Adobe TIFF technical note 3; libtiff fpAcc, which the reference reads such
           * files through, lib/iio/iio.c:1463-1661): every row is stored as byte planes - the most significant
           * bytes of all its samples first - differenced byte-wise with a stride of one pixel's samples */
          const size_t ns = (size_t)cw * cpp, rb = ns * bytes;
          unsigned char *tmp = malloc(rb ? rb : 1);
          if (!tmp) { free(out); free(raw); return fail(path, "out of memory"); }
          for (uint64_t r = 0; r < rows_in; ++r) {
            unsigned char *row = raw + r * rb;
            for (size_t i = cpp; i < rb; ++i) row[i] = (unsigned char)(row[i] + row[i - cpp]);
            memcpy(tmp, row, rb);
            for (size_t j = 0; j < ns; ++j)
              for (int k = 0; k < bytes; ++k) /* k = significance, 0 = least */
                row[j * bytes + (t.le ? k : bytes - 1 - k)] = tmp[(size_t)(bytes - 1 - k) * ns + j];
          }
          free(tmp);
        }
        if (pred == 2 && bytes <= 4 && fmt != 3) /* horizontal differencing */
          for (uint64_t r = 0; r < rows_in; ++r)
            for (uint64_t x = cpp; x < cw * cpp; ++x) {
              unsigned char *p = raw + (r * cw * cpp + x) * bytes, *q = p - cpp * bytes;
              if (bytes == 1) p[0] += q[0];
              else {
                uint32_t a = 0, c2 = 0;
                for (int k = 0; k < bytes; ++k) { a |= (uint32_t)p[t.le ? k : bytes - 1 - k] << (8 * k); c2 |= (uint32_t)q[t.le ? k : bytes - 1 - k] << (8 * k); }
                a += c2;
                for (int k = 0; k < bytes; ++k) p[t.le ? k : bytes - 1 - k] = (unsigned char)(a >> (8 * k));
              }
            }
        for (uint64_t r = 0; r < rows; ++r)
          for (uint64_t x = 0; x < cols; ++x)
            for (uint64_t c = 0; c < cpp; ++c) {
              const unsigned char *p = raw + ((r * cw + x) * cpp + c) * bytes;
              uint64_t v = 0;
              for (int k = 0; k < bytes; ++k) v |= (uint64_t)p[t.le ? k : bytes - 1 - k] << (8 * k);
              float f;
              if (fmt == 3) {
                if (bytes == 4) { uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); }
                else if (bytes == 8) { double dd; memcpy(&dd, &v, 8); f = (float)dd; }
                else { free(out); free(raw); return fail(path, "unsupported float sample size"); }
              } else if (fmt == 2) {
                f = bytes == 1 ? (float)(int8_t)v : bytes == 2 ? (float)(int16_t)v : bytes == 4 ? (float)(int32_t)v : (float)(int64_t)v;
              } else f = (float)v;
              out[((cy * cl + r) * W + cx * cw + x) * spp + (planar == 2 ? pl : c)] = f;
            }
      }
  free(raw);
  *w = (int)W; *h = (int)H; *ch = (int)spp;
  return out;
}

/* -------------------------------------------------------------------- PNG */
static uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }
static float *read_png(const char *path, const unsigned char *b, size_t n, int *w, int *h, int *ch) {
  size_t pos = 8, zlen = 0;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0, npal = 0;
  unsigned char *z = malloc(n ? n : 1), pal[256 * 3];
  if (!z) return fail(path, "out of memory");
  memset(pal, 0, sizeof pal);
  while (pos + 12 <= n) {
    const uint32_t len = be32(b + pos);
    const unsigned char *ty = b + pos + 4, *dat = b + pos + 8;
    if (pos + 12 + len > n) break;
    if (!memcmp(ty, "IHDR", 4) && len >= 13) { W = be32(dat); H = be32(dat + 4); depth = dat[8]; ctype = dat[9]; interlace = dat[12]; }
    else if (!memcmp(ty, "PLTE", 4)) { npal = len / 3; memcpy(pal, dat, len > 768 ? 768 : len); }
    else if (!memcmp(ty, "IDAT", 4)) { memcpy(z + zlen, dat, len); zlen += len; }
    else if (!memcmp(ty, "IEND", 4)) break;
    pos += 12 + len;
  }
  (void)npal;
  if (!W || !H || interlace || (depth != 8 && depth != 16)) { free(z); return fail(path, "unsupported PNG (need 8/16-bit, non-interlaced)"); }
  if (!sane_size(W, H, 4) || (ctype != 0 && ctype != 2 && ctype != 3 && ctype != 4 && ctype != 6) || (ctype == 3 && depth != 8)) {
    free(z);
    return fail(path, "bad PNG header");
  }
  const int comps = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
  const int bpp = comps * depth / 8;
  const size_t stride = (size_t)W * bpp;
  size_t rawlen = (stride + 1) * H;
  unsigned char *raw = malloc(rawlen);
  if (!raw) { free(z); return fail(path, "out of memory"); }
  if (z_uncompress(raw, &rawlen, z, zlen) || rawlen < (stride + 1) * H) { free(z); free(raw); return fail(path, "PNG inflate failed"); }
  free(z);
  for (uint32_t y = 0; y < H; ++y) { /* undo the per-row filters */
    unsigned char *cur = raw + y * (stride + 1) + 1, *up = y ? cur - (stride + 1) : NULL;
    const int ft = cur[-1];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, bb = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
      int add = 0;
      if (ft == 1) add = a;
      else if (ft == 2) add = bb;
      else if (ft == 3) add = (a + bb) / 2;
      else if (ft == 4) { const int p = a + bb - c, pa = abs(p - a), pb = abs(p - bb), pc = abs(p - c); add = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); }
      cur[i] = (unsigned char)(cur[i] + add);
    }
  }
  const int och = ctype == 3 ? 3 : comps;
  float *out = malloc((size_t)W * H * och * sizeof(float));
  if (!out) { free(raw); return fail(path, "out of memory"); }
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = 0; x < W; ++x)
      for (int c = 0; c < och; ++c) {
        const unsigned char *p = raw + y * (stride + 1) + 1 + (size_t)x * bpp;
        float v;
        if (ctype == 3) v = pal[p[0] * 3 + c];
        else v = depth == 8 ? p[c] : (float)(p[2 * c] << 8 | p[2 * c + 1]);
        out[((size_t)y * W + x) * och + c] = v;
      }
  free(raw);
  *w = (int)W; *h = (int)H; *ch = och;
  return out;
}

float *img_read(const char *path, int *w, int *h, int *ch) {
  if (!path) return NULL;
  size_t n;
  unsigned char *b = slurp(path, &n);
  if (!b) return fail(path, "cannot open");
  float *d = NULL;
  if (n >= 12 && !memcmp(b, "PIEH", 4)) d = read_flo(path, b, n, w, h, ch);
  else if (n >= 8 && ((b[0] == 'I' && b[1] == 'I') || (b[0] == 'M' && b[1] == 'M'))) d = read_tiff(path, b, n, w, h, ch);
  else if (n >= 8 && !memcmp(b, "\x89PNG\r\n\x1a\n", 8)) d = read_png(path, b, n, w, h, ch);
  else if (n >= 8 && b[0] == 'P' && (b[1] == 'f' || b[1] == 'F')) d = read_pfm(path, b, n, w, h, ch);
  else if (n >= 8 && b[0] == 'P' && (b[1] == '2' || b[1] == '3' || b[1] == '5' || b[1] == '6')) d = read_pnm(path, b, n, w, h, ch);
  else if (n >= 4 && b[0] == 0xFF && b[1] == 0xD8) d = nlk_read_jpeg(path, b, n, w, h, ch);
  else fail(path, "unknown image format (supported: TIFF, PNG, JPEG, PFM, PGM / PPM, FLO)");
  free(b);
  return d;
}

/* ------------------------------------------------------------------ write */
static void put(unsigned char **p, uint64_t v, int bytes) { for (int i = 0; i < bytes; ++i) *(*p)++ = (unsigned char)(v >> (8 * i)); }

static int write_tiff(const char *path, const float *d, int w, int h, int ch) {
  const size_t cnt = (size_t)w * h * ch;
  int as_bytes = 1; /* the reference's writer stores 8 bits when every sample is an integer in [0,255] */
  for (size_t i = 0; i < cnt && as_bytes; ++i) as_bytes = d[i] >= 0 && d[i] <= 255 && d[i] == floorf(d[i]);
  const int bytes = as_bytes ? 1 : 4;
  uint64_t datalen = (uint64_t)cnt * bytes;
  /* The reference's writer (libtiff, lib/iio/iio.c:3022-3026) LZW-compresses images below 4 Mpixel. Float
   * noise does not compress and the encoder would dominate a frame's time, so files are written
   * uncompressed by default (every reader concerned takes both); NLK_TIFF_LZW=1 selects LZW, one strip. */
  const char *want_lzw = getenv("NLK_TIFF_LZW");
  const int lzw = want_lzw && atoi(want_lzw) && (uint64_t)w * h < 2000ull * 2000ull;
  unsigned char *packed = NULL;
  if (lzw) {
    const size_t cap = (size_t)datalen + datalen / 2 + 64;
    unsigned char *src = (unsigned char *)d, *tmp = NULL;
    if (as_bytes) {
      tmp = malloc(cnt ? cnt : 1);
      if (!tmp) return -1;
      for (size_t i = 0; i < cnt; ++i) tmp[i] = (unsigned char)d[i];
      src = tmp;
    }
    packed = malloc(cap);
    if (!packed) { free(tmp); return -1; }
    const size_t plen = lzw_encode(src, (size_t)datalen, packed, cap);
    free(tmp);
    if (plen > cap) { free(packed); return -1; }
    datalen = plen;
  }
  const int big = datalen > 0xF0000000ull;
  FILE *f = fopen(path, "wb");
  if (!f) { free(packed); return -1; }
  unsigned char hdr[512], *p = hdr;
  const int nent = 10, esz = big ? 20 : 12;
  const uint64_t ifd_off = big ? 16 : 8;
  const uint64_t bps_off = ifd_off + (big ? 8 : 2) + (uint64_t)nent * esz + (big ? 8 : 4);
  const uint64_t data_off = bps_off + 16;
  put(&p, 0x4949, 2);
  if (big) { put(&p, 43, 2); put(&p, 8, 2); put(&p, 0, 2); put(&p, ifd_off, 8); put(&p, nent, 8); }
  else { put(&p, 42, 2); put(&p, ifd_off, 4); put(&p, nent, 2); }
#define ENT(tag, type, count, val) do { put(&p, tag, 2); put(&p, type, 2); put(&p, count, big ? 8 : 4); put(&p, val, big ? 8 : 4); } while (0)
  const int ltype = big ? 16 : 4;
  ENT(256, 4, 1, (uint64_t)w);
  ENT(257, 4, 1, (uint64_t)h);
  if (ch <= (big ? 4 : 2)) { uint64_t v = 0; for (int c = 0; c < ch; ++c) v |= (uint64_t)(8 * bytes) << (16 * c); ENT(258, 3, (uint64_t)ch, v); }
  else ENT(258, 3, (uint64_t)ch, bps_off);
  ENT(259, 3, 1, lzw ? 5 : 1);              /* LZW on request, else none */
  ENT(262, 3, 1, ch >= 3 ? 2 : 1);          /* RGB / min-is-black (reference: lib/iio/iio.c:3000-3020) */
  ENT(273, ltype, 1, data_off);
  ENT(277, 3, 1, (uint64_t)ch);
  ENT(278, 4, 1, (uint64_t)h);
  ENT(279, ltype, 1, datalen);
  ENT(339, 3, 1, as_bytes ? 1 : 3);         /* sample format: uint / IEEE float */
  put(&p, 0, big ? 8 : 4);                  /* no next IFD */
  for (int c = 0; c < 8; ++c) put(&p, c < ch ? 8 * bytes : 0, 2);
  fwrite(hdr, 1, (size_t)data_off, f);
  if (packed) { fwrite(packed, 1, (size_t)datalen, f); free(packed); }
  else if (as_bytes) { for (size_t i = 0; i < cnt; ++i) fputc((int)d[i], f); }
  else fwrite(d, 4, cnt, f);
  return fclose(f);
}

static __thread uint32_t crc_table[256];
static uint32_t crc32_upd(uint32_t c, const unsigned char *b, size_t n) {
  if (!crc_table[1]) for (uint32_t i = 0; i < 256; ++i) { uint32_t k = i; for (int j = 0; j < 8; ++j) k = k & 1 ? 0xEDB88320u ^ (k >> 1) : k >> 1; crc_table[i] = k; }
  c = ~c;
  for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ b[i]) & 255] ^ (c >> 8);
  return ~c;
}
static void png_chunk(FILE *f, const char *ty, const unsigned char *d, uint32_t n) {
  unsigned char l[4] = {(unsigned char)(n >> 24), (unsigned char)(n >> 16), (unsigned char)(n >> 8), (unsigned char)n};
  fwrite(l, 1, 4, f); fwrite(ty, 1, 4, f); if (n) fwrite(d, 1, n, f);
  uint32_t c = crc32_upd(crc32_upd(0, (const unsigned char *)ty, 4), d, n);
  unsigned char cc[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
  fwrite(cc, 1, 4, f);
}
static int write_png(const char *path, const float *d, int w, int h, int ch) {
  if (ch < 1 || ch > 4) return -1;
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  static const int ct[5] = {0, 0, 4, 2, 6};
  const size_t stride = (size_t)w * ch + 1, rawn = stride * h;
  /* zlib stream of stored (uncompressed) deflate blocks */
  unsigned char *raw = malloc(rawn), *z = malloc(rawn + rawn / 65535 * 5 + 16), *q = z;
  for (int y = 0; y < h; ++y) {
    raw[y * stride] = 0;
    for (size_t i = 0; i < (size_t)w * ch; ++i) { float v = d[(size_t)y * w * ch + i]; raw[y * stride + 1 + i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
  }
  *q++ = 0x78; *q++ = 0x01;
  uint32_t a = 1, b2 = 0;
  for (size_t o = 0; o < rawn;) {
    const size_t k = rawn - o > 65535 ? 65535 : rawn - o;
    *q++ = o + k == rawn; *q++ = k & 255; *q++ = k >> 8; *q++ = ~k & 255; *q++ = (~k >> 8) & 255;
    memcpy(q, raw + o, k); q += k;
    for (size_t i = 0; i < k; ++i) { a = (a + raw[o + i]) % 65521; b2 = (b2 + a) % 65521; }
    o += k;
  }
  const uint32_t ad = b2 << 16 | a;
  *q++ = ad >> 24; *q++ = ad >> 16; *q++ = ad >> 8; *q++ = ad;
  fwrite("\x89PNG\r\n\x1a\n", 1, 8, f);
  unsigned char ih[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
                          (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h,
                          8, (unsigned char)ct[ch], 0, 0, 0};
  png_chunk(f, "IHDR", ih, 13);
  png_chunk(f, "IDAT", z, (uint32_t)(q - z));
  png_chunk(f, "IEND", NULL, 0);
  free(raw); free(z);
  return fclose(f);
}

int img_write(const char *path, const float *d, int w, int h, int ch) {
  const char *ext = strrchr(path, '.');
  if (ext && (!strcasecmp(ext, ".tif") || !strcasecmp(ext, ".tiff"))) return write_tiff(path, d, w, h, ch);
  if (ext && !strcasecmp(ext, ".png")) return write_png(path, d, w, h, ch);
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  if (ext && !strcasecmp(ext, ".flo") && ch == 2) {
    const int32_t wh[2] = {w, h};
    fwrite("PIEH", 1, 4, f);
    fwrite(wh, 4, 2, f);
  } else if (ext && !strcasecmp(ext, ".pfm") && (ch == 1 || ch == 3)) {
    fprintf(f, "P%c\n%d %d\n-1\n", ch == 3 ? 'F' : 'f', w, h);
  } else {
    fclose(f);
    fprintf(stderr, "imgio: %s: cannot write %d channels in this format (use .tif)\n", path, ch);
    return -1;
  }
  fwrite(d, 4, (size_t)w * h * ch, f);
  return fclose(f);
}
