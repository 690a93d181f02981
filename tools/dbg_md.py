import importlib, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("bwd-nlkalman_amd")
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    w, h, ch, sigma = 320, 256, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 5)
    o0, o1 = pkg.rgb2opp(n0), pkg.rgb2opp(n1)
    p1, p2 = (pkg.default_params(sigma, m) for m in (pkg.FLT1, pkg.FLT2))
    f0 = np.load("/tmp/f0.npy") if os.path.exists("/tmp/f0.npy") else pkg.filter_frame(o0, None, None, sigma, p1)
    np.save("/tmp/f0.npy", f0)
    hole = f0.copy(); hole[100:130, 50:90] = np.nan
    f1 = np.load("/tmp/f1.npy") if os.path.exists("/tmp/f1.npy") else pkg.filter_frame(o1, hole, None, sigma, p1)
    np.save("/tmp/f1.npy", f1)
    f2 = pkg.filter_frame(o1, hole, f1, sigma, p2)
    f2b = pkg.filter_frame(o1, f0, f1, sigma, p2)
    f2c = pkg.filter_frame(o1, None, f1, sigma, p2)
    np.savez(sys.argv[2], f2=f2, f2b=f2b, f2c=f2c)
else:
    for tag, env in (("one", {}), ("split", {"NLK_DEVICES": "0,0"})):
        subprocess.run([sys.executable, __file__, "child", f"/tmp/{tag}.npz"], env=dict(os.environ, **env), check=True)
    a, b = np.load("/tmp/one.npz"), np.load("/tmp/split.npz")
    for k in ("f2", "f2b", "f2c"):
        d = np.abs(a[k] - b[k])
        ys, xs, cs = np.nonzero(d > 5e-4)
        print(k, "max", d.max(), "count", len(ys), "rows", np.unique(ys)[:40], "cols", np.unique(xs)[:30])
