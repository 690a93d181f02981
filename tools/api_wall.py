"""Wall time of the host-pointer drop-in API (PCIe + per-call allocations included)
and of the command-line tool on a 1080p frame; for DESIGN.md §5. Run with gpurun."""
import importlib, os, subprocess, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = 1920, 1080, 3, 20.0
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
p = pkg.default_params(sigma, pkg.FLT1)
o0, o1 = pkg.rgb2opp(n0), pkg.rgb2opp(n1)
prev = pkg.filter_frame(o0, None, None, sigma, p)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = pkg.filter_frame(o1, prev, None, sigma, p); ts.append(time.perf_counter() - t0)
print("nlkalman_filter_frame (host pointers) ms:", [round(t * 1e3, 2) for t in ts], "-> Mpix/s", round(w * h / min(ts) / 1e6, 1))
if os.environ.get("NLK_API_WALL_ONLY"):
    sys.exit(0)
with tempfile.TemporaryDirectory() as d:
    def wpfm(path, a):
        with open(path, "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (a.shape[1], a.shape[0])); f.write(np.ascontiguousarray(a, np.float32).tobytes())
    wpfm(d + "/n1.pfm", n1); wpfm(d + "/p.pfm", pkg.opp2rgb(prev))
    exe = os.path.join(ROOT, "bwd-nlkalman_amd", "bin", "nlkalman-flt")
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.check_call([exe, "-i", d + "/n1.pfm", "-s", "20", "--flt10", d + "/p.pfm", "--flt11", d + "/o.tif", "--f2_p", "0"])
        print("nlkalman-flt process wall s:", round(time.perf_counter() - t0, 3))
