#define _GNU_SOURCE /* struct ucred, SO_PEERCRED, MSG_CMSG_CLOEXEC */
/* cli_server.c — see cli_server.h. Wire format of a request, client -> server on a unix stream socket:
 *   [u32 payload bytes] with the client's stdout and stderr attached (SCM_RIGHTS), then the payload:
 *   "NLK1" 0, tool 0, working directory 0, argc (decimal) 0, argv[0] 0 ... argv[argc-1] 0
 * and the answer: [i32 exit status]. */
#include "cli_server.h"

#include <errno.h>
#include <limits.h>
#include <setjmp.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>

#include "nlk_hip.h"

/* ---- what a tool may not do by itself inside a resident process */
static jmp_buf g_back;
static int g_serving = 0;
static volatile int g_status = 0;

void cli_warm_join(void); /* cli_args.c */

void cli_exit(int status) {
  cli_warm_join(); /* nobody leaves - exit() and its handlers, or the jump back - while the warm-up thread is in hipInit */
  if (!g_serving) exit(status);
  g_status = status;
  longjmp(g_back, 1);
}

static struct cli_dev { struct nlk_ctx *c; void *p; } *g_dev = NULL;
static int g_ndev = 0, g_dev_cap = 0;

int cli_dev_alloc(struct nlk_ctx *c, void **p, size_t bytes) {
  if (g_serving && g_ndev == g_dev_cap) { /* (an allocation that cannot be tracked is not made: it would never be released) */
    const int cap = g_dev_cap ? 2 * g_dev_cap : 64;
    struct cli_dev *t = (struct cli_dev *)realloc(g_dev, (size_t)cap * sizeof *t);
    if (!t) return NLK_ENOMEM;
    g_dev = t;
    g_dev_cap = cap;
  }
  const int rc = nlk_dev_alloc(c, p, bytes);
  if (rc == NLK_OK && g_serving) {
    g_dev[g_ndev].c = c;
    g_dev[g_ndev++].p = *p;
  }
  return rc;
}

int cli_dev_free(struct nlk_ctx *c, void *p) {
  for (int i = 0; i < g_ndev; ++i)
    if (g_dev[i].p == p) {
      g_dev[i] = g_dev[--g_ndev];
      break;
    }
  return nlk_dev_free(c, p);
}

/* host buffers of a request (images read, staging): one release at the request's end, whichever way the tool left */
static void **g_host = NULL;
static int g_nhost = 0, g_host_cap = 0;

void *cli_host_keep(void *p) {
  if (!p) return NULL;
  if (g_nhost == g_host_cap) {
    const int cap = g_host_cap ? 2 * g_host_cap : 32;
    void **t = (void **)realloc(g_host, (size_t)cap * sizeof *t);
    if (!t) { free(p); return NULL; }
    g_host = t;
    g_host_cap = cap;
  }
  g_host[g_nhost++] = p;
  return p;
}

void cli_host_release(void) {
  for (int i = 0; i < g_nhost; ++i) free(g_host[i]);
  g_nhost = 0;
}

void cli_dev_release(void) {
  for (int i = 0; i < g_ndev; ++i) {
    (void)nlk_sync(g_dev[i].c);
    (void)nlk_dev_free(g_dev[i].c, g_dev[i].p);
  }
  g_ndev = 0;
}

/* ---- both ends of the socket */
#define CLI_MAX_FDS 8               /* descriptors a request may carry before it is refused (2 are expected) */
#define CLI_SERVER_IO_TIMEOUT_S 10  /* a connected client must send its request within this */
#define CLI_CLIENT_IO_TIMEOUT_S 600 /* a client waits this long for the answer (a 4K frame call takes ~0.1 s) */
static int write_all(int fd, const void *buf, size_t n) {
  const char *p = (const char *)buf;
  while (n) {
    const ssize_t k = write(fd, p, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return -1;
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

static int read_all(int fd, void *buf, size_t n) {
  char *p = (char *)buf;
  while (n) {
    const ssize_t k = read(fd, p, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return -1;
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

static int unix_address(const char *path, struct sockaddr_un *a) {
  memset(a, 0, sizeof *a);
  a->sun_family = AF_UNIX;
  if (strlen(path) >= sizeof a->sun_path) return -1;
  strcpy(a->sun_path, path);
  return 0;
}

int cli_remote(const char *tool, int argc, const char **argv) {
  const char *path = getenv("NLK_SERVER");
  struct sockaddr_un a;
  if (!path || !path[0] || unix_address(path, &a)) return -1;
  const int s = socket(AF_UNIX, SOCK_STREAM, 0);
  if (s < 0) return -1;
  if (connect(s, (struct sockaddr *)&a, sizeof a)) {
    close(s);
    return -1; /* nobody listening: the caller does the work itself */
  }
  {
    /* a request may take as long as a frame call does - but not for ever (a wedged server) */
    const struct timeval tv = {CLI_CLIENT_IO_TIMEOUT_S, 0};
    setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    setsockopt(s, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
  }
  char cwd[PATH_MAX];
  if (!getcwd(cwd, sizeof cwd)) strcpy(cwd, ".");
  size_t n = 5 + strlen(tool) + 1 + strlen(cwd) + 1 + 16;
  for (int i = 0; i < argc; ++i) n += strlen(argv[i]) + 1;
  char *buf = (char *)malloc(n), *q = buf;
  if (!buf) { close(s); return -1; }
  q += sprintf(q, "NLK1") + 1;
  q += sprintf(q, "%s", tool) + 1;
  q += sprintf(q, "%s", cwd) + 1;
  q += sprintf(q, "%d", argc) + 1;
  for (int i = 0; i < argc; ++i) q += sprintf(q, "%s", argv[i]) + 1;
  const uint32_t len = (uint32_t)(q - buf);

  /* the length word travels with this process's stdout and stderr */
  fflush(stdout);
  fflush(stderr);
  struct msghdr m;
  struct iovec io = {(void *)&len, sizeof len};
  union { char b[CMSG_SPACE(2 * sizeof(int))]; struct cmsghdr align; } ctl;
  memset(&m, 0, sizeof m);
  memset(&ctl, 0, sizeof ctl);
  m.msg_iov = &io;
  m.msg_iovlen = 1;
  m.msg_control = ctl.b;
  m.msg_controllen = sizeof ctl.b;
  struct cmsghdr *cm = CMSG_FIRSTHDR(&m);
  cm->cmsg_level = SOL_SOCKET;
  cm->cmsg_type = SCM_RIGHTS;
  cm->cmsg_len = CMSG_LEN(2 * sizeof(int));
  const int fds[2] = {1, 2};
  memcpy(CMSG_DATA(cm), fds, sizeof fds);
  int32_t status = -1;
  const int sent = sendmsg(s, &m, 0) == (ssize_t)sizeof len && !write_all(s, buf, len);
  free(buf);
  if (!sent || read_all(s, &status, sizeof status)) {
    /* the request was (perhaps) taken and the answer never came: running the tool again here could write the
       outputs twice - report instead */
    fprintf(stderr, "%s: the server at %s went away\n", tool, path);
    close(s);
    return 1;
  }
  close(s);
  return (int)(status & 0xFF);
}

int cli_serve(const char *path, const struct cli_tool *tools) {
  struct sockaddr_un a;
  if (unix_address(path, &a)) return fprintf(stderr, "nlk-server: socket path too long\n"), 1;
  /* a server already answering on this path keeps it (an unconditional unlink would silently take its socket over) */
  {
    const int probe = socket(AF_UNIX, SOCK_STREAM, 0);
    if (probe >= 0) {
      const int live = connect(probe, (struct sockaddr *)&a, sizeof a) == 0;
      close(probe);
      if (live) return fprintf(stderr, "nlk-server: another server is listening on %s\n", path), 1;
    }
  }
  const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
  if (ls < 0) return perror("nlk-server: socket"), 1;
  unlink(path); /* (a stale socket file nobody listens on) */
  /* the socket belongs to this user alone: whoever can connect makes the server read and write image files under
     its uid in a directory of the client's choosing. Mode 0600 from the start (umask around bind), and every
     connection's peer must be this uid (SO_PEERCRED below). */
  const mode_t um = umask(0177);
  const int bound = bind(ls, (struct sockaddr *)&a, sizeof a);
  umask(um);
  if (bound || chmod(path, 0600) || listen(ls, 64)) return perror("nlk-server: bind / listen"), 1;
  signal(SIGPIPE, SIG_IGN);
  char home[PATH_MAX];
  if (!getcwd(home, sizeof home)) strcpy(home, "/");

  for (;;) {
    const int cs = accept(ls, NULL, NULL);
    if (cs < 0) {
      if (errno == EINTR) continue;
      perror("nlk-server: accept");
      break;
    }
    /* the requests are served one after the other: a client that connects and then says nothing must not hold
       every pipeline step up for ever */
    {
      const struct timeval tv = {CLI_SERVER_IO_TIMEOUT_S, 0};
      setsockopt(cs, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
      setsockopt(cs, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
      struct ucred cr;
      socklen_t crl = sizeof cr;
      if (getsockopt(cs, SOL_SOCKET, SO_PEERCRED, &cr, &crl) || cr.uid != geteuid()) {
        close(cs);
        continue;
      }
    }
    /* length word + the client's stdout / stderr */
    uint32_t len = 0;
    struct msghdr m;
    struct iovec io = {&len, sizeof len};
    union { char b[CMSG_SPACE(CLI_MAX_FDS * sizeof(int))]; struct cmsghdr align; } ctl;
    memset(&m, 0, sizeof m);
    m.msg_iov = &io;
    m.msg_iovlen = 1;
    m.msg_control = ctl.b;
    m.msg_controllen = sizeof ctl.b;
    int fds[2] = {-1, -1};
    const ssize_t got = recvmsg(cs, &m, MSG_WAITALL | MSG_CMSG_CLOEXEC);
    /* every descriptor that arrived is ours to close: exactly two are expected; any other count, other control
       messages' descriptors, or a truncated control buffer (more were sent than fit) end the request */
    int nfd = 0, bad_fds = (m.msg_flags & MSG_CTRUNC) != 0;
    int all[CLI_MAX_FDS];
    if (got >= 0)
      for (struct cmsghdr *cm = CMSG_FIRSTHDR(&m); cm; cm = CMSG_NXTHDR(&m, cm))
        if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS) {
          const int k = (int)((cm->cmsg_len - CMSG_LEN(0)) / sizeof(int));
          for (int i = 0; i < k; ++i) {
            int fd;
            memcpy(&fd, CMSG_DATA(cm) + i * sizeof(int), sizeof fd);
            if (nfd < CLI_MAX_FDS) all[nfd++] = fd;
            else { close(fd); bad_fds = 1; }
          }
        }
    if (nfd == 2 && !bad_fds) { fds[0] = all[0]; fds[1] = all[1]; }
    else {
      for (int i = 0; i < nfd; ++i) close(all[i]);
      if (nfd) bad_fds = 1;
    }
    if (got != (ssize_t)sizeof len || len < 8 || len > (1u << 20) || bad_fds) {
      if (fds[0] >= 0) close(fds[0]);
      if (fds[1] >= 0) close(fds[1]);
      close(cs);
      continue;
    }
    char *buf = (char *)malloc((size_t)len + 1);
    if (!buf || read_all(cs, buf, len)) {
      free(buf);
      if (fds[0] >= 0) close(fds[0]);
      if (fds[1] >= 0) close(fds[1]);
      close(cs);
      continue;
    }
    buf[len] = 0;
    /* payload -> tool, directory, argv (every string must end inside the payload: buf[len] is a NUL of ours) */
    const char *end = buf + len, *q = buf;
#define NEXT_STRING(var)                         \
  const char *var = NULL;                        \
  if (q && q < end) { var = q; q += strlen(q) + 1; } else q = NULL;
    NEXT_STRING(magic)
    NEXT_STRING(tool)
    NEXT_STRING(cwd)
    NEXT_STRING(argc_s)
    const int argc = argc_s ? atoi(argc_s) : -1;
    int ok = magic && tool && cwd && argc_s && !strcmp(magic, "NLK1") && argc >= 0 && argc < 4096;
    const char **argv = ok ? (const char **)calloc((size_t)argc + 1, sizeof(char *)) : NULL;
    ok = ok && argv;
    for (int i = 0; ok && i < argc; ++i) {
      NEXT_STRING(a)
      if (!a) ok = 0;
      else argv[i] = a;
    }
#undef NEXT_STRING
    int32_t status = 125;
    int stop = 0;
    if (ok && !strcmp(tool, "shutdown")) {
      status = 0;
      stop = 1;
    } else if (ok) {
      cli_tool_fn fn = NULL;
      for (const struct cli_tool *t = tools; t->name; ++t)
        if (!strcmp(t->name, tool)) fn = t->fn;
      /* the tool writes to the CLIENT's stdout / stderr and works in the client's directory */
      fflush(stdout);
      fflush(stderr);
      const int so = dup(1), se = dup(2);
      if (fds[0] >= 0) dup2(fds[0], 1);
      if (fds[1] >= 0) dup2(fds[1], 2);
      if (!fn) {
        fprintf(stderr, "nlk-server: no tool `%s` here\n", tool);
        status = 127;
      } else if (chdir(cwd)) {
        fprintf(stderr, "nlk-server: cannot enter %s\n", cwd);
        status = 126;
      } else {
        g_serving = 1;
        if (setjmp(g_back) == 0) status = fn(argc, argv);
        else status = g_status;
        g_serving = 0;
        cli_dev_release();
        cli_host_release();
        cli_trace_reset();
      }
      fflush(stdout);
      fflush(stderr);
      dup2(so, 1);
      dup2(se, 2);
      close(so);
      close(se);
      if (chdir(home)) { /* (the next request brings its own directory) */ }
    }
    if (fds[0] >= 0) close(fds[0]);
    if (fds[1] >= 0) close(fds[1]);
    (void)write_all(cs, &status, sizeof status);
    close(cs);
    free((void *)argv);
    free(buf);
    if (stop) break;
  }
  close(ls);
  unlink(path);
  return 0;
}
