/* main_multiscale.c — `decompose`, `recompose`, `merge_coarse`: drop-ins for the multiscale
 * wrapper's tools (reference: lib/multiscale/decompose.cpp, recompose.cpp, merge_coarse.cpp,
 * option picking multiscaler.cpp:109-123), which scripts/msnlkalman-seq.sh runs around the
 * filter. One source; the tool is chosen by the program name. Same arguments:
 *
 *   decompose    input prefix levels suffix [-r ratio]     level i -> <prefix><i><suffix>
 *   recompose    prefix levels suffix output [-c factor]
 *   merge_coarse image coarse result [-c factor]
 *
 * The whole-image DCTs run on the GPU (nlk_dev_image_dct: two matrix products per channel on the
 * matrix cores); images stay resident between the transforms and the coefficient copies. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cli_server.h"
#include "imgio.h"
#include "nlk_hip.h"

nlk_ctx *nlkalman_hip_context(void);

static nlk_ctx *C;
#define CHK(call)                                                          \
  do {                                                                     \
    if ((call) != NLK_OK) {                                                \
      fprintf(stderr, "multiscale (hip): %s\n", nlk_last_error(C));        \
      cli_exit(EXIT_FAILURE);                                                  \
    }                                                                      \
  } while (0)

/* "-o value" anywhere on the command line (removed from argv); flags (dflt == NULL) return
 * non-NULL when present (reference: multiscaler.cpp:109-123) */
static const char *pick_option(int *argc, char **argv, const char *opt, const char *dflt) {
  const int has_value = dflt != NULL;
  for (int i = 1; i < *argc - has_value; ++i)
    if (argv[i][0] == '-' && !strcmp(argv[i] + 1, opt)) {
      const char *r = has_value ? argv[i + 1] : argv[i];
      for (int j = i; j < *argc - has_value - 1; ++j) argv[j] = argv[j + has_value + 1];
      *argc -= has_value + 1;
      return r;
    }
  return dflt;
}

struct dimg {
  float *d;
  int w, h, ch;
};

static struct dimg load(const char *path) {
  struct dimg im;
  float *host = img_read(path, &im.w, &im.h, &im.ch);
  if (!host) cli_exit(EXIT_FAILURE);
  void *d = NULL;
  const size_t bytes = (size_t)im.w * im.h * im.ch * sizeof(float);
  CHK(cli_dev_alloc(C, &d, bytes));
  CHK(nlk_h2d(C, d, host, bytes));
  free(host);
  im.d = (float *)d;
  return im;
}

static void save(const char *path, const float *d, int w, int h, int ch) {
  const size_t bytes = (size_t)w * h * ch * sizeof(float);
  float *host = malloc(bytes);
  CHK(nlk_d2h(C, host, d, bytes));
  if (img_write(path, host, w, h, ch)) { fprintf(stderr, "cannot write %s\n", path); cli_exit(EXIT_FAILURE); }
  free(host);
}

static int decompose(int argc, char **argv) {
  const float ratio = atof(pick_option(&argc, argv, "r", "2."));
  const int usage = pick_option(&argc, argv, "h", NULL) != NULL;
  if (argc != 5 || usage) {
    fprintf(stderr, "Usage: %s input prefix levels suffix [-r ratio]\n", argv[0]);
    return EXIT_FAILURE;
  }
  const int levels = atoi(argv[3]);
  C = nlkalman_hip_context();
  struct dimg im = load(argv[1]);
  CHK(nlk_dev_image_dct(C, im.d, im.w, im.h, im.ch, 0));
  int w = im.w, h = im.h;
  void *lvl = NULL;
  CHK(cli_dev_alloc(C, &lvl, (size_t)im.w * im.h * im.ch * sizeof(float)));
  for (int i = 0; i < levels; ++i) {  /* decompose.cpp:31-56 */
    if (w < 1 || h < 1) { fprintf(stderr, "decompose: level %d is empty\n", i); return EXIT_FAILURE; }
    CHK(nlk_dev_copy_block(C, (float *)lvl, w, im.d, im.w, im.ch, w, h));
    CHK(nlk_dev_image_dct(C, (float *)lvl, w, h, im.ch, 1));
    char name[2048];
    snprintf(name, sizeof name, "%s%d%s", argv[2], i, argv[4]);
    save(name, (float *)lvl, w, h, im.ch);
    w /= ratio;
    h /= ratio;
  }
  return EXIT_SUCCESS;
}

/* the first rows*factor x cols*factor coefficients of `from` replace those of `into` */
static void low_frequencies(struct dimg into, struct dimg from, float factor) {
  int bh = 0, bw = 0;  /* loop bounds of recompose.cpp:43-44: j < rows * factor in float */
  while (bh < from.h * factor) ++bh;
  while (bw < from.w * factor) ++bw;
  if (bh > into.h) bh = into.h;
  if (bw > into.w) bw = into.w;
  CHK(nlk_dev_copy_block(C, into.d, into.w, from.d, from.w, from.ch, bw, bh));
}

static int recompose(int argc, char **argv) {
  const float factor = atof(pick_option(&argc, argv, "c", ".8"));
  const int usage = pick_option(&argc, argv, "h", NULL) != NULL;
  if (argc != 5 || usage) {
    fprintf(stderr, "Usage: %s prefix levels suffix output [-c factor]\n", argv[0]);
    return EXIT_FAILURE;
  }
  const int levels = atoi(argv[2]);
  C = nlkalman_hip_context();
  char name[2048];
  snprintf(name, sizeof name, "%s0%s", argv[1], argv[3]);
  struct dimg out = load(name);
  CHK(nlk_dev_image_dct(C, out.d, out.w, out.h, out.ch, 0));
  for (int i = 1; i < levels; ++i) {
    snprintf(name, sizeof name, "%s%d%s", argv[1], i, argv[3]);
    struct dimg im = load(name);
    CHK(nlk_dev_image_dct(C, im.d, im.w, im.h, im.ch, 0));
    low_frequencies(out, im, factor);
    CHK(nlk_sync(C));
    cli_dev_free(C, im.d);
  }
  CHK(nlk_dev_image_dct(C, out.d, out.w, out.h, out.ch, 1));
  save(argv[4], out.d, out.w, out.h, out.ch);
  return EXIT_SUCCESS;
}

static int merge_coarse(int argc, char **argv) {
  const float factor = atof(pick_option(&argc, argv, "c", ".8"));
  const int usage = pick_option(&argc, argv, "h", NULL) != NULL;
  if (argc != 4 || usage) {
    fprintf(stderr, "Usage: %s image coarse result [-c factor]\n", argv[0]);
    return EXIT_FAILURE;
  }
  C = nlkalman_hip_context();
  struct dimg fine = load(argv[1]);
  CHK(nlk_dev_image_dct(C, fine.d, fine.w, fine.h, fine.ch, 0));
  struct dimg coarse = load(argv[2]);
  CHK(nlk_dev_image_dct(C, coarse.d, coarse.w, coarse.h, coarse.ch, 0));
  low_frequencies(fine, coarse, factor);
  CHK(nlk_dev_image_dct(C, fine.d, fine.w, fine.h, fine.ch, 1));
  save(argv[3], fine.d, fine.w, fine.h, fine.ch);
  return EXIT_SUCCESS;
}

/* the three tools as one function that looks at the name it is called by: main() below, or the resident server */
int nlk_tool_multiscale(int argc, const char **argv_c) {
  char **argv = (char **)argv_c; /* (the options are removed from the pointer array; the strings stay as they are) */
  const char *base = strrchr(argv[0], '/');
  base = base ? base + 1 : argv[0];
  if (!strcmp(base, "decompose")) return decompose(argc, argv);
  if (!strcmp(base, "recompose")) return recompose(argc, argv);
  if (!strcmp(base, "merge_coarse")) return merge_coarse(argc, argv);
  fprintf(stderr, "%s: call me as decompose, recompose or merge_coarse\n", base);
  return EXIT_FAILURE;
}

#ifndef NLK_TOOL_NO_MAIN
int main(int argc, const char **argv) {
  const char *base = strrchr(argv[0], '/');
  const int remote = cli_remote(base ? base + 1 : argv[0], argc, argv); /* a resident server (NLK_SERVER), if there is one */
  return remote >= 0 ? remote : nlk_tool_multiscale(argc, argv);
}
#endif
