"""Development aid: per-section wave cycles of k_group8m (needs a build with -DNLK_EXP=4)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = 1920, 1080, 3, 20.0
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 7)
ctx = pkg.Context(0)
p = pkg.default_params(sigma, pkg.FLT1)
d0, d1 = ctx.upload(n0), ctx.upload(n1)
o0, o1 = ctx.alloc(n0.nbytes), ctx.alloc(n0.nbytes)
ctx.rgb2opp(d0, w, h, ch); ctx.rgb2opp(d1, w, h, ch)
ctx.filter_frame(o0, d0, None, None, w, h, ch, sigma, p)
ctx.sync()
L = pkg.hip() if hasattr(pkg, "hip") else pkg._hip
out = (C.c_ulonglong * 8)()
L.nlk_debug_read(out, 1)
ctx.filter_frame(o1, d1, o0, None, w, h, ch, sigma, p)
ctx.sync()
L.nlk_debug_read(out, 1)
v = np.array(list(out), dtype=np.float64)
names = ["prologue", "passA jobs", "gains", "A->B", "passB", "init+flush", "targets", "total"]
nt = v[6]
for n, x in zip(names, v):
    print(f"{n:12s} {x:14.0f}  per target {x/nt:10.1f}")
