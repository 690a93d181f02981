"""Timing of the TV-L1 flow at one size (development aid; run with gpurun).
   python tools/tvl1_time.py [w h [fscale [lam]]]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
a = sys.argv[1:]
w, h = (int(a[0]), int(a[1])) if len(a) >= 2 else (1920, 1080)
fscale = int(a[2]) if len(a) >= 3 else 0
lam = float(a[3]) if len(a) >= 4 else 0.4
n0, n1, _ = synth.noisy_pair(w, h, 3, 20.0, 7)
ctx = pkg.Context(0)
d0, d1 = ctx.upload(n0), ctx.upload(n1)
g0, g1 = ctx.alloc(w * h * 4), ctx.alloc(w * h * 4)
ctx.gray(g0, d0, w, h, 3); ctx.gray(g1, d1, w, h, 3)
d_f = ctx.alloc(w * h * 8)
p = pkg.tvl1_params(w, h, lam=lam, fscale=fscale)
it = ctx.tvl1_flow(d_f, g0, g1, w, h, p); ctx.sync()
for batch in (os.environ.get("NLK_TV_BATCH", "12"),):
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        it = ctx.tvl1_flow(d_f, g0, g1, w, h, p)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{w}x{h} fscale {fscale} lambda {lam}: {p.nscales} scales, {it} iterations, {dt*1e3:.2f} ms "
          f"({dt/it*1e6:.1f} us per iteration, {w*h/dt/1e6:.1f} Mpix/s)")
f = ctx.download(d_f, (h, w, 2))
print("flow median", np.median(f[..., 0]), np.median(f[..., 1]), "finite", np.isfinite(f).all())
