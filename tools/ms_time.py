import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("bwd-nlkalman_amd")
ctx = pkg.Context(0)
for (w, h) in [(1920, 1080), (3840, 2160)]:
    a = np.random.default_rng(0).uniform(0, 255, (h, w, 3)).astype(np.float32)
    d = ctx.upload(a)
    ctx.image_dct(d, w, h, 3, False); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.image_dct(d, w, h, 3, False)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    fl = 2.0 * 3 * (h * h * w + h * w * w)
    print(f"{w}x{h}x3 DCT: {dt*1e3:.2f} ms, {fl/dt/1e12:.1f} TFLOP/s")
    ctx.free(d)
