import importlib, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
for ch in (1, 3):
    w, h, sigma = 1920, 1080, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 7)
    ctx = pkg.Context(0)
    d0 = ctx.upload(n0)
    o0 = ctx.alloc(n0.nbytes)
    p1 = pkg.default_params(sigma, pkg.FLT1)
    for rep in range(3):
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            ctx.filter_frame(o0, d0, None, None, w, h, ch, sigma, p1)
            ctx.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("ch", ch, "per-call wall ms (sync after each):", [round(t, 2) for t in ts])
