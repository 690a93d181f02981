/* nlk-imgconv — read any supported image (TIFF / PNG / PFM / FLO) and write it
 * by extension; the counterpart of the reference tool chain's `iion`
 * (lib/imscript-lite/src/iion.c) for the formats of host/imgio.c. Host only. */
#include <stdio.h>
#include <stdlib.h>

#include "imgio.h"

int main(int argc, char **argv) {
  if (argc != 3) return fprintf(stderr, "usage: %s in out\n", argv[0]), 2;
  int w, h, ch;
  float *d = img_read(argv[1], &w, &h, &ch);
  if (!d) return 1;
  if (img_write(argv[2], d, w, h, ch)) return fprintf(stderr, "cannot write %s\n", argv[2]), 1;
  printf("%d %d %d\n", w, h, ch);
  free(d);
  return 0;
}
