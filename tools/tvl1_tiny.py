"""Per-iteration cost of the one-workgroup TV-L1 level (k_tv_level_wg): one-scale flows on tiny images.
   python tools/tvl1_tiny.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
ctx = pkg.Context(0)
for w, h in ((15, 8), (30, 17), (60, 34)):
    n0, n1, _ = synth.noisy_pair(w * 8, h * 8, 3, 20.0, 7)
    g0 = np.ascontiguousarray(n0[::8, ::8, 1][:h, :w]); g1 = np.ascontiguousarray(n1[::8, ::8, 1][:h, :w])
    d0, d1 = ctx.upload(g0), ctx.upload(g1)
    d_f = ctx.alloc(w * h * 8)
    p = pkg.tvl1_params(w, h, lam=0.25)
    p.nscales = 1
    for rep in range(2):
        it = ctx.tvl1_flow(d_f, d0, d1, w, h, p); ctx.sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.L.nlk_dev_tvl1_flow(ctx.h, d_f, d0, d1, w, h, __import__("ctypes").byref(p), None)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{w}x{h}: {it} iterations, {dt*1e6:.1f} us per flow, {dt/max(it,1)*1e6:.2f} us per iteration (warps and launch included)")
