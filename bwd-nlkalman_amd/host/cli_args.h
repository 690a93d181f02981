/* cli_args.h — command-line parser with the flag grammar the reference's CLIs
 * expose through lib/argparse (reference: lib/argparse/argparse.c:80-105 value
 * parsing, :138-187 option matching, :210-268 parse loop, :276-366 usage text):
 *   --name value | --name=value | -x value | -xvalue, integers via strtol(.., 0),
 *   floats via strtof, last occurrence wins, `--` ends the options, non-option
 *   arguments are ignored, unknown option => message + usage + exit(1),
 *   -h/--help => usage + exit(0).
 * Own implementation (table driven); nothing of lib/argparse is reused. */
#ifndef NLK_CLI_ARGS_H
#define NLK_CLI_ARGS_H

enum cli_type { CLI_END, CLI_GROUP, CLI_STRING, CLI_INT, CLI_FLOAT };

struct cli_option {
  enum cli_type type;
  char short_name;        /* 0 = none */
  const char *long_name;  /* NULL for a group header: `help` is the title */
  void *value;            /* const char** / int* / float* */
  const char *help;
};

/* parses argv (skipping argv[0]); exits on -h, unknown options and bad values */
void cli_parse(const struct cli_option *opts, const char *prog, const char *description,
               int argc, const char **argv);
void cli_usage(const struct cli_option *opts, const char *prog, const char *description);

/* One frame call per process is how the pipelines use the tools (scripts/nlkalman-seq.sh:39-41): the ~0.1 s the
 * HIP runtime takes to come up is on every call's critical path. cli_warm_start() begins creating the
 * process-wide device context on a second thread as soon as the arguments are known good, so that it overlaps
 * the reading / decoding of the input files; cli_warm_join() waits for it (call it before the first use of the
 * context). cli_trace() prints the elapsed time since program start to stderr when NLK_CLI_TRACE is set. */
void cli_warm_start(void);
void cli_warm_join(void);
void cli_trace(const char *what);
/* last trace point; returns `status`. (Leaving through _exit() instead of the exit handlers was measured: no
 * gain - the ~0.3 s of a one-shot process are the dynamic loading of the ROCm runtime before main (~0.1 s) and
 * hipInit (~0.17 s), against ~0.05 s of reading, computing and writing: profiles/README.md round 4.) */
int cli_leave(int status);

#endif
