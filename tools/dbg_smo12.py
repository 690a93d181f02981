"""debug: where does the 12x12 smoother's NaN pattern differ from the oracle (case rgb84x60_p12_s40)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases, oracle as O
pkg = importlib.import_module("bwd-nlkalman_amd")
name = "rgb84x60_p12_s40"
ref = cases.run_chain(O, name)
I = cases.inputs(name)
s, over = I["sigma"], I["over"]
ps = pkg.default_params(s, pkg.SMO1, **{k: v for k, v in over.items() if k != "search_sz_x"})
po = O.Params(*[getattr(ps, k) for k, _ in ps._fields_])
r, tr = O.smooth_frame(ref["f2_0"], ref["ws"], None, s, po, trace=True)
g = pkg.smooth_frame(ref["f2_0"], ref["ws"], None, s, ps)
ng, nr = np.isnan(g), np.isnan(r)
print("nan ours", ng.sum(), "nan ref", nr.sum(), "ws nan", np.isnan(ref["ws"]).sum())
d = np.abs(np.where(ng | nr, 0, g - r))
print("maxabs", d.max(), "count >2e-3", (d > 2e-3).sum())
ys, xs, cs = np.nonzero(ng != nr)
print("positions", list(zip(ys[:20], xs[:20], cs[:20])))
print("np0 hist", np.bincount(tr["np0"][tr["active"].astype(bool)])[:10], "nagg", np.unique(tr["nagg"]))
os.environ["NLK_GROUP12_ROWS"] = "1"
g2 = pkg.smooth_frame(ref["f2_0"], ref["ws"], None, s, ps)
print("old kernel: nan", np.isnan(g2).sum(), "maxabs", np.abs(np.where(np.isnan(g2) | nr, 0, g2 - r)).max())
