"""Where a one-shot process (the command-line tools: one frame call per process, scripts/nlkalman-seq.sh:39-41)
spends its start-up: library load, context creation (HIP runtime + device), the first frame call (code objects
loaded on first launch), a second call. Run with gpurun.   python tools/startup_times.py [w h]"""
import ctypes as C, importlib, os, subprocess, sys, tempfile, time
t_start = time.perf_counter()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
pkg = importlib.import_module("bwd-nlkalman_amd")
L = pkg.hip()
t1 = time.perf_counter()
ctx = pkg.Context(0)
t2 = time.perf_counter()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
synth = importlib.import_module("bwd-nlkalman_amd.synth")
n0, n1, _ = synth.noisy_pair(w, h, 3, 20.0, 1)
p = pkg.default_params(20.0, pkg.FLT1)
d0, d1, o = ctx.upload(n0), ctx.upload(n1), ctx.alloc(n0.nbytes)
ctx.sync()
t3 = time.perf_counter()
ctx.filter_frame(o, d1, d0, None, w, h, 3, 20.0, p); ctx.sync()
t4 = time.perf_counter()
ctx.filter_frame(o, d1, d0, None, w, h, 3, 20.0, p); ctx.sync()
t5 = time.perf_counter()
print(f"numpy import {t0 - t_start:.3f} s | dlopen libnlk_hip.so {t1 - t0:.3f} | nlk_ctx_create {t2 - t1:.3f} | uploads {t3 - t2:.3f} | "
      f"first temporal frame call {t4 - t3:.3f} | second {t5 - t4:.4f}")
exe = os.path.join(ROOT, "bwd-nlkalman_amd", "bin", "nlkalman-flt")
with tempfile.TemporaryDirectory() as d:
    def wpfm(path, a):
        with open(path, "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (a.shape[1], a.shape[0])); f.write(np.ascontiguousarray(a, np.float32).tobytes())
    wpfm(d + "/n1.pfm", n1); wpfm(d + "/p.pfm", n0)
    for env in ({}, {"NLK_CLI_TRACE": "1"}):
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            r = subprocess.run([exe, "-i", d + "/n1.pfm", "-s", "20", "--flt10", d + "/p.pfm", "--flt11", d + "/o.pfm", "--f2_p", "0"],
                               env=dict(os.environ, **env), capture_output=True, text=True)
            ts.append(time.perf_counter() - t)
        print("nlkalman-flt wall s", env, [round(x, 3) for x in ts], r.stderr[-900:].replace("\n", " | "))
    t = time.perf_counter(); subprocess.run([exe, "-h"], capture_output=True); print("nlkalman-flt -h wall s", round(time.perf_counter() - t, 3))
    # the same call with bin/nlk-server holding the device (host/cli_server.h): NLK_SERVER=<socket>
    srv = os.path.join(ROOT, "bwd-nlkalman_amd", "bin", "nlk-server")
    sock = d + "/nlk.sock"
    t = time.perf_counter()
    proc = subprocess.Popen([srv, sock], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    while not os.path.exists(sock) and proc.poll() is None:
        time.sleep(0.01)
    print("nlk-server ready after s", round(time.perf_counter() - t, 3))
    for env in ({}, {"NLK_CLI_TRACE": "1"}):
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            r = subprocess.run([exe, "-i", d + "/n1.pfm", "-s", "20", "--flt10", d + "/p.pfm", "--flt11", d + "/o2.pfm", "--f2_p", "0"],
                               env=dict(os.environ, NLK_SERVER=sock, **env), capture_output=True, text=True)
            ts.append(time.perf_counter() - t)
        print("nlkalman-flt through nlk-server wall s", env, [round(x, 3) for x in ts], r.returncode, r.stderr[-300:].replace("\n", " | "))
    subprocess.run([srv, "--stop", sock], capture_output=True); proc.wait(timeout=30)
