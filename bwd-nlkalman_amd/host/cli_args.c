/* cli_args.c — see cli_args.h. */
#include "cli_args.h"
#include "cli_server.h" /* cli_exit: leaves the process, or goes back to the resident server's loop */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void die_opt(const struct cli_option *o, int is_long, const char *reason) {
  if (is_long) fprintf(stderr, "error: option `--%s` %s\n", o->long_name, reason);
  else fprintf(stderr, "error: option `-%c` %s\n", o->short_name, reason);
  cli_exit(1);
}

/* `inl` = value glued to the option (--x=v, -xv) or NULL: then the next argv is used */
static void take_value(const struct cli_option *o, int is_long, const char *inl, int *i, int argc,
                       const char **argv) {
  const char *v = inl;
  if (!v) {
    if (*i + 1 >= argc) die_opt(o, is_long, "requires a value");
    v = argv[++*i];
  }
  char *end = NULL;
  switch (o->type) {
    case CLI_STRING:
      *(const char **)o->value = v;
      break;
    case CLI_INT:
      *(int *)o->value = (int)strtol(v, &end, 0);
      if (*end) die_opt(o, is_long, "expects an integer value");
      break;
    case CLI_FLOAT:
      *(float *)o->value = strtof(v, &end);
      if (*end) die_opt(o, is_long, "expects a numerical value");
      break;
    default:
      break;
  }
}

void cli_usage(const struct cli_option *opts, const char *prog, const char *description) {
  printf("Usage: %s [options] [[--] args]\n   or: %s [options]\n", prog, prog);
  if (description) printf("\n%s\n", description);
  printf("\n    -h, --help            show this help message and exit\n");
  for (const struct cli_option *o = opts; o->type != CLI_END; ++o) {
    if (o->type == CLI_GROUP) {
      printf("\n%s\n", o->help);
      continue;
    }
    char head[64];
    const char *ty = o->type == CLI_STRING ? "<str>" : (o->type == CLI_INT ? "<int>" : "<flt>");
    int n = 0;
    if (o->short_name) n += snprintf(head + n, sizeof head - n, "-%c, ", o->short_name);
    snprintf(head + n, sizeof head - n, "--%s=%s", o->long_name, ty);
    /* option column is 22 wide after a 4-space indent; longer heads wrap */
    if (strlen(head) <= 20) printf("    %-22s%s\n", head, o->help);
    else printf("    %s\n%26s%s\n", head, "", o->help);
  }
  printf("\n");
}

void cli_parse(const struct cli_option *opts, const char *prog, const char *description,
               int argc, const char **argv) {
  for (int i = 1; i < argc; ++i) {
    const char *a = argv[i];
    if (a[0] != '-' || !a[1]) continue; /* not an option: ignored */
    if (a[1] != '-') {                  /* short option, value glued or next */
      if (a[1] == 'h' && !a[2]) {
        cli_usage(opts, prog, description);
        cli_exit(0);
      }
      const struct cli_option *o = opts;
      for (; o->type != CLI_END; ++o)
        if (o->type != CLI_GROUP && o->short_name && o->short_name == a[1]) break;
      if (o->type == CLI_END) goto unknown;
      take_value(o, 0, a[2] ? a + 2 : NULL, &i, argc, argv);
      continue;
    }
    if (!a[2]) break; /* `--` */
    if (!strcmp(a + 2, "help")) {
      cli_usage(opts, prog, description);
      cli_exit(0);
    }
    {
      const struct cli_option *o = opts;
      const char *inl = NULL;
      for (; o->type != CLI_END; ++o) {
        if (o->type == CLI_GROUP || !o->long_name) continue;
        const size_t n = strlen(o->long_name);
        if (strncmp(a + 2, o->long_name, n)) continue;
        if (a[2 + n] == '\0') { inl = NULL; break; }
        if (a[2 + n] == '=') { inl = a + 2 + n + 1; break; }
      }
      if (o->type == CLI_END) goto unknown;
      take_value(o, 1, inl, &i, argc, argv);
      continue;
    }
  unknown:
    fprintf(stderr, "error: unknown option `%s`\n", a);
    cli_usage(opts, prog, description);
    cli_exit(1);
  }
}

/* ---- start-up helpers of the command-line tools (cli_args.h) */
#include <pthread.h>
#include <time.h>

int nlkalman_hip_context_warm(void); /* libnlkalman.so: creates the process-wide context if it can; never exits */

static pthread_t g_warm_thread;
static int g_warm_started = 0;

/* (a failure is not this thread's to report: the tool's own first use of the context tries again, prints and exits -
 * one thread in exit(), after cli_warm_join()) */
static void *warm_main(void *arg) {
  (void)arg;
  (void)nlkalman_hip_context_warm();
  return NULL;
}

void cli_warm_start(void) {
  if (!g_warm_started && pthread_create(&g_warm_thread, NULL, warm_main, NULL) == 0) g_warm_started = 1;
}

void cli_warm_join(void) {
  if (g_warm_started) pthread_join(g_warm_thread, NULL);
  g_warm_started = 0;
}

static int g_trace_on = -1;
static struct timespec g_trace_t0;

void cli_trace_reset(void) { /* (resident server: a request's trace counts from the request, not from the server's start) */
  if (g_trace_on > 0) clock_gettime(CLOCK_MONOTONIC, &g_trace_t0);
}

void cli_trace(const char *what) {
  struct timespec t;
  if (g_trace_on < 0) {
    g_trace_on = getenv("NLK_CLI_TRACE") != NULL;
    clock_gettime(CLOCK_MONOTONIC, &g_trace_t0);
  }
  const int on = g_trace_on;
  const struct timespec t0 = g_trace_t0;
  if (!on) return;
  clock_gettime(CLOCK_MONOTONIC, &t);
  fprintf(stderr, "[cli %8.2f ms] %s\n", (t.tv_sec - t0.tv_sec) * 1e3 + (t.tv_nsec - t0.tv_nsec) * 1e-6, what);
}

int cli_leave(int status) {
  cli_trace("leaving");
  return status;
}
