"""Per-kernel times of the four frame calls at one size (development aid; run with gpurun).
   python tools/mode_times.py [w h ch sigma [reps]]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
a = sys.argv[1:]
w, h, ch = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1920, 1080, 3)
sigma = float(a[3]) if len(a) >= 4 else 20.0
reps = int(a[4]) if len(a) >= 5 else 10
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 7)
ctx = pkg.Context(0)
d0, d1 = ctx.upload(n0), ctx.upload(n1)
ctx.rgb2opp(d0, w, h, ch); ctx.rgb2opp(d1, w, h, ch)
o0, o1, o2, o3 = (ctx.alloc(n0.nbytes) for _ in range(4))
p1, p2, p3 = (pkg.default_params(sigma, m) for m in (pkg.FLT1, pkg.FLT2, pkg.SMO1))
calls = {
    "FLT1 spatial": lambda: ctx.filter_frame(o0, d0, None, None, w, h, ch, sigma, p1),
    "FLT1 temporal": lambda: ctx.filter_frame(o1, d1, o0, None, w, h, ch, sigma, p1),
    "FLT2 temporal": lambda: ctx.filter_frame(o2, d1, o0, o1, w, h, ch, sigma, p2),
    "SMO1": lambda: ctx.smooth_frame(o3, o0, o2, None, w, h, ch, sigma, p3),
}
import time
for name, fn in calls.items():
    fn(); ctx.sync()
    t0 = time.perf_counter()   # wall time without the profiler (the banded two-stream pipeline runs only then)
    for _ in range(reps):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.set_profiling(True)
    for _ in range(reps):
        fn()
    ctx.sync()
    tm = ctx.timings()
    ctx.set_profiling(False)
    print(f"{name:14s} " + "  ".join(f"{k[:-3]} {v:.3f}" for k, v in tm.items()) + f"  wall {wall:.3f}")
