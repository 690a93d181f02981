/* cli_server.h — the one-frame-per-process tools behind a resident process.
 *
 * The reference's pipelines start a process per step and frame (scripts/nlkalman-seq.sh:39-41, 80-81, 100-102:
 * tvl1flow, nlkalman-flt twice, nlkalman-smo). A HIP process costs ~0.27 s before its first kernel (dynamic
 * loading of the runtime + hipInit: profiles/README.md round 4) against ~10 ms of work per step. With
 *
 *     bin/nlk-server /tmp/nlk.sock &        export NLK_SERVER=/tmp/nlk.sock
 *
 * the SAME command lines - the script runs unchanged - are carried out by the server, which holds the device
 * context, the loaded code objects and the scratch buffers: a tool started with NLK_SERVER set connects to that
 * unix socket, hands over its arguments, its working directory and its own stdout / stderr (descriptor passing),
 * and exits with the status the server returns. No server listening: the tool does the work itself, as before.
 * Requests are served one at a time, in order of arrival.
 *
 * A tool's entry point becomes a function; what it may not do in a resident process goes through this header:
 * leave with cli_exit() instead of exit(), allocate device memory with cli_dev_alloc() (released after the
 * request whatever path the tool left by). */
#ifndef NLK_CLI_SERVER_H
#define NLK_CLI_SERVER_H

#include <stddef.h>

struct nlk_ctx;
typedef int (*cli_tool_fn)(int argc, const char **argv);
struct cli_tool {
  const char *name; /* NULL ends the table */
  cli_tool_fn fn;
};

/* In a tool's main(): the exit status the server returned, or -1 when there is no server to ask (NLK_SERVER
 * unset / nobody listening): run the tool in this process then. */
int cli_remote(const char *tool, int argc, const char **argv);
/* The server loop on the unix socket `path` (created; an existing file of that name is replaced). Returns when a
 * client asks for the tool "shutdown" (bin/nlk-server --stop <socket>). */
int cli_serve(const char *path, const struct cli_tool *tools);
/* exit(status) - or, inside the server, back to its loop with that status */
void cli_exit(int status);
/* host memory of a request: cli_host_keep(p) registers a malloc'ed buffer (returns p; NULL stays NULL) and
 * cli_host_release() frees every registered one - the tools' wrappers and the server's loop call it, so that an
 * early `return 1` or a cli_exit() from deep inside leaks nothing in a resident process */
void *cli_host_keep(void *p);
void cli_host_release(void);
void cli_trace_reset(void); /* cli_args.c: the elapsed-time trace starts again with the next request */
/* nlk_dev_alloc, remembered; cli_dev_release frees what the current request allocated (a one-shot process never
 * needs to) */
int cli_dev_alloc(struct nlk_ctx *c, void **p, size_t bytes);
int cli_dev_free(struct nlk_ctx *c, void *p); /* nlk_dev_free of something cli_dev_alloc returned */
void cli_dev_release(void);

#endif
