"""Hostile clients against the request loop of bin/nlk-server (host/cli_server.c), CPU only: the loop is built
with AddressSanitizer + UBSan around a stand-in tool table (no HIP), then 2000 connections that do not speak
the protocol - nothing, random bytes, absurd lengths, mutated payloads, impossible argument counts, directories
that do not exist - interleaved with well-formed requests that must keep being answered.
   python tools/fuzz_cli_server.py"""
import array, os, random, socket, struct, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = r'''
#include <stdio.h>
#include <string.h>
#include "cli_server.h"
#include "nlk_hip.h"
int nlk_dev_alloc(nlk_ctx *c, void **p, size_t n) { (void)c; (void)p; (void)n; return 1; }
int nlk_dev_free(nlk_ctx *c, void *p) { (void)c; (void)p; return 0; }
int nlk_sync(nlk_ctx *c) { (void)c; return 0; }
static int echo_tool(int argc, const char **argv) {
  for (int i = 0; i < argc; ++i) printf("[%s]", argv[i]);
  printf("\n");
  if (argc > 1 && !strcmp(argv[1], "die")) cli_exit(7);
  return argc;
}
int main(int argc, char **argv) {
  static const struct cli_tool tools[] = {{"echo", echo_tool}, {NULL, NULL}};
  (void)argc;
  return cli_serve(argv[1], tools);
}
'''

with tempfile.TemporaryDirectory() as d:
    open(d + "/stub.c", "w").write(STUB)
    host = os.path.join(ROOT, "bwd-nlkalman_amd", "host")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu99",
                           "-I" + host, "-I" + os.path.join(ROOT, "include"), "-o", d + "/srv", d + "/stub.c",
                           os.path.join(host, "cli_server.c")])
    path = d + "/s.sock"
    log = open(d + "/srv.err", "w")   # (a file: the loop reports every dropped client, a pipe would fill up)
    srv = subprocess.Popen([d + "/srv", path], stdout=log, stderr=log,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1"))
    while not os.path.exists(path):
        time.sleep(0.02)

    def good(args):
        payload = b"NLK1\0echo\0/tmp\0" + str(len(args)).encode() + b"\0" + b"".join(a + b"\0" for a in args)
        c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        c.connect(path)
        r, w = os.pipe()
        c.sendmsg([struct.pack("<I", len(payload))], [(socket.SOL_SOCKET, socket.SCM_RIGHTS, array.array("i", [w, w]))])
        c.sendall(payload)
        st = c.recv(4)
        c.close()
        os.close(w)
        out = os.read(r, 65536)
        os.close(r)
        return struct.unpack("<i", st)[0], out

    assert good([b"echo", b"a", b"bb"]) == (3, b"[echo][a][bb]\n")
    assert good([b"echo", b"die"])[0] == 7          # a tool that leaves through cli_exit
    random.seed(4)
    for it in range(2000):
        kind = random.randrange(6)
        c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        c.connect(path)
        try:
            if kind == 1:
                c.sendall(os.urandom(random.randrange(1, 64)))
            elif kind == 2:
                ln = random.choice([0, 1, 7, 8, 9, 100, 1 << 20, (1 << 20) + 1, 0xffffffff])
                c.sendall(struct.pack("<I", ln) + os.urandom(random.randrange(0, 200)))
            elif kind == 3:
                p = bytearray(b"NLK1\0echo\0/tmp\0" + b"3\0a\0b\0c\0")
                for _ in range(random.randrange(1, 4)):
                    p[random.randrange(len(p))] = random.randrange(256)
                c.sendall(struct.pack("<I", max(len(p) + random.choice([0, 0, 0, -3, 5]), 0)) + bytes(p))
            elif kind == 4:
                p = b"NLK1\0echo\0/tmp\0" + str(random.choice([-1, 0, 5, 4095, 4096, 10 ** 9])).encode() + b"\0x\0"
                c.sendall(struct.pack("<I", len(p)) + p)
            elif kind == 5:
                p = b"NLK1\0echo\0/nonexistent_dir\0" + b"1\0x\0"
                c.sendall(struct.pack("<I", len(p)) + p)
            c.settimeout(0.02)
            try:
                c.recv(4)
            except Exception:
                pass
        except (BrokenPipeError, ConnectionResetError):
            pass
        c.close()
        if it % 250 == 0:
            assert good([b"echo", b"still", b"here"]) == (3, b"[echo][still][here]\n"), it
    assert srv.poll() is None, open(d + "/srv.err", errors="replace").read()[-2000:]
    srv.kill()
    srv.wait()
    log.close()
    err = open(d + "/srv.err", errors="replace").read()
    assert "ERROR: AddressSanitizer" not in err and "runtime error" not in err, err[-2000:]
    print("2000 hostile connections: the loop kept answering; no sanitizer report")
