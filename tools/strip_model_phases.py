"""Per-phase device times of ONE middle rank of a world of N stepped alone (exchanges skipped: csrc/strips.hip dry
run) - where a rank's step goes at that world size. Run with gpurun.   python tools/strip_model_phases.py [N ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = 1920, 1080, 3, 20.0
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
ctx = pkg.Context(0)
p = pkg.default_params(sigma, pkg.FLT1)
d0, d1, dp = ctx.upload(pkg.rgb2opp(n0)), ctx.upload(pkg.rgb2opp(n1)), ctx.alloc(n0.nbytes)
ctx.filter_frame(dp, d0, None, None, w, h, ch, sigma, p); ctx.sync()
for nw in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    m = pkg.Strips([0], nw // 2, nw, w, h, ch, sigma, p)
    m.set_dry_run(True)
    m.load(0, d1, dp)
    for _ in range(5):
        m.step()
    m.sync()
    m.set_options(overlap=os.environ.get('OVERLAP') == '1', timing=True, graph=False)
    for _ in range(30):
        m.step()
    ph, us, _ = m.stats()
    print(f"world {nw}: rows {m.geometry(0)['gy1'] - m.geometry(0)['gy0']}", ph, "sum", round(sum(ph.values()), 4))
    m.close()
