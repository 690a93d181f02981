/* nlk-server — the one-frame-per-process tools of the pipelines behind ONE resident process (cli_server.h).
 *
 *   nlk-server [--lazy] <socket>     serve nlkalman-flt, nlkalman-smo, tvl1flow and the multiscale tools
 *                                    (decompose, recompose, merge_coarse) on that unix socket
 *   nlk-server --stop <socket>       ask the server there to leave
 *
 * The tools find it through NLK_SERVER=<socket>; scripts/nlkalman-seq.sh (reference: :39-41, 80-81, 100-102) runs
 * unchanged and pays the ~0.27 s start of a HIP process once instead of four times per frame. --lazy: the device
 * context is created by the first request that needs it instead of at start-up. The server's own environment
 * (NLK_* switches, NLK_DEVICES) is what counts, not the clients'. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cli_server.h"

struct nlk_ctx;
struct nlk_ctx *nlkalman_hip_context(void); /* libnlkalman.so: the process-wide device context */
int nlk_tool_flt(int argc, const char **argv);
int nlk_tool_smo(int argc, const char **argv);
int nlk_tool_tvl1(int argc, const char **argv);
int nlk_tool_multiscale(int argc, const char **argv);

int main(int argc, const char **argv) {
  int lazy = 0, stop = 0;
  const char *path = NULL;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--lazy")) lazy = 1;
    else if (!strcmp(argv[i], "--stop")) stop = 1;
    else path = argv[i];
  }
  if (!path) return fprintf(stderr, "usage: nlk-server [--lazy] <socket> | nlk-server --stop <socket>\n"), 1;
  if (stop) {
    setenv("NLK_SERVER", path, 1);
    const int rc = cli_remote("shutdown", 0, NULL);
    return rc < 0 ? (fprintf(stderr, "nlk-server: nobody listens at %s\n", path), 1) : rc;
  }
  static const struct cli_tool tools[] = {
      {"nlkalman-flt", nlk_tool_flt}, {"nlkalman-smo", nlk_tool_smo}, {"tvl1flow", nlk_tool_tvl1},
      {"decompose", nlk_tool_multiscale}, {"recompose", nlk_tool_multiscale}, {"merge_coarse", nlk_tool_multiscale},
      {NULL, NULL}};
  if (!lazy) (void)nlkalman_hip_context();
  return cli_serve(path, tools);
}
