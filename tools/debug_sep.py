"""Debug aid: every stage of the two-frame chain with the separable group kernel (default) against the Kronecker one
(NLK_GROUP_SEP=0), stage by stage on the SAME inputs (the Kronecker kernel's outputs feed both).
  python tools/debug_sep.py [case ...]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cases  # noqa: E402

B = importlib.import_module("bwd-nlkalman_amd")


def run(sep, fn):
    os.environ["NLK_GROUP_SEP"] = str(sep)
    B.reload_switches()
    return fn()


for name in (sys.argv[1:] or list(cases.CASES)):
    if cases.CASES[name][4].get("patch_sz", 8) != 8:
        continue
    ref = run(0, lambda: cases.run_chain(B, name))
    for sep in (2, 6):
        got = run(sep, lambda: cases.run_chain_stagewise(B, ref, name))
        worst = 0.0
        for k in ("f1_0", "f2_0", "f1_1", "f2_1", "s1_0"):
            a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
            nan_a, nan_b = np.isnan(a), np.isnan(b)
            d = np.abs(np.where(nan_a | nan_b, 0, a - b))
            worst = max(worst, d.max())
            if d.max() > 1e-3 or nan_a.sum() != nan_b.sum():
                print(f"{name:18s} sep={sep} {k:5s} nan {int(nan_a.sum()):6d} / {int(nan_b.sum()):6d}  max|d| {d.max():.3e}  rmse {np.sqrt((d ** 2).mean()):.3e}  n>1e-2 {int((d > 1e-2).sum())}")
        print(f"{name:18s} sep={sep} worst max|d| {worst:.3e}")
