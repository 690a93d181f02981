"""Per-kernel times of FLT1 temporal at one size for a list of patch sizes (development aid; run with gpurun).
   python tools/psz_times.py w h ch sigma psz [psz ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 7)
ctx = pkg.Context(0)
d0, d1 = ctx.upload(n0), ctx.upload(n1)
ctx.rgb2opp(d0, w, h, ch); ctx.rgb2opp(d1, w, h, ch)
o0, o1 = ctx.alloc(n0.nbytes), ctx.alloc(n0.nbytes)
for psz in map(int, sys.argv[5:]):
    p = pkg.default_params(sigma, pkg.FLT1, patch_sz=psz)
    ctx.filter_frame(o0, d0, None, None, w, h, ch, sigma, p)
    ctx.filter_frame(o1, d1, o0, None, w, h, ch, sigma, p); ctx.sync()
    ctx.set_profiling(True)
    for _ in range(5):
        ctx.filter_frame(o1, d1, o0, None, w, h, ch, sigma, p)
    ctx.sync()
    tm = ctx.timings()
    ctx.set_profiling(False)
    print(f"psz {psz:2d}: " + "  ".join(f"{k[:-3]} {v:.3f}" for k, v in tm.items()))
