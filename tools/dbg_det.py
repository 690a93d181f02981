"""debug: run-to-run equality of the deterministic mode, many repetitions"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = 200, 136, 3, 20.0
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 17)
o0, o1 = pkg.rgb2opp(n0), pkg.rgb2opp(n1)
c = pkg.Context(0)
c.set_deterministic(True)
def frame(cur, prev, basic, p, smo=False):
    d = [c.upload(a) if a is not None else None for a in (cur, prev, basic)]
    o = c.alloc(cur.nbytes)
    (c.smooth_frame if smo else c.filter_frame)(o, d[0], d[1], d[2], w, h, ch, sigma, p)
    out = c.download(o, cur.shape)
    for x in d + [o]:
        if x: c.free(x)
    return out
for psz in [int(a) for a in sys.argv[1:]] or [6, 12, 10]:
    p1 = pkg.default_params(sigma, pkg.FLT1, patch_sz=psz, search_sz_x=min(10, 3 * (psz // 2)))
    ref0 = frame(o0, None, None, p1)
    hole = ref0.copy(); hole[40:70, 100:160] = np.nan
    ref1 = frame(o1, hole, None, p1)
    bad0 = bad1 = 0
    for it in range(30):
        a = frame(o0, None, None, p1); b = frame(o1, hole, None, p1)
        if not np.array_equal(a, ref0):
            bad0 += 1; d = np.abs(a - ref0); print("  spatial diff", psz, it, d.max(), (d > 0).sum(), np.argwhere(d.max(axis=2) > 0)[:3].tolist())
        if not np.array_equal(b, ref1, equal_nan=True):
            bad1 += 1; d = np.abs(np.nan_to_num(b - ref1)); print("  temporal diff", psz, it, d.max(), (d > 0).sum(), np.argwhere(d.max(axis=2) > 0)[:3].tolist())
    print("psz", psz, "spatial mismatches", bad0, "temporal mismatches", bad1, "of 30")
