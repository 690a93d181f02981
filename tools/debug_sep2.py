import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cases
B = importlib.import_module("bwd-nlkalman_amd")
I = cases.inputs("gray64_s20")
p1 = B.default_params(I["sigma"], 0)
o0 = B.rgb2opp(I["n0"])
for kron in (True, False):
    if kron: os.environ["NLK_GROUP_KRON"] = "1"
    else: os.environ.pop("NLK_GROUP_KRON", None)
    B.reload_switches()
    out = B.filter_frame(o0, None, None, I["sigma"], p1)
    sys.stdout.flush()
