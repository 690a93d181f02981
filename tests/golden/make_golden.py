"""Writes tests/golden/<case>.npz: expected outputs of every pipeline stage for
the seeded cases of tests/cases.py, produced by the CPU oracle (serial order).

    python tests/golden/make_golden.py

Provenance: these vectors come from oracle/nlk_oracle.c, NOT from a build of
the reference (it cannot be built here: FFTW3 is absent, see oracle/nlk_oracle.c).
They pin the oracle against regressions and give the GPU tests a fixed target
that does not depend on the oracle library being rebuilt identically.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import oracle as O  # noqa: E402

KEEP = ("f1_0", "f2_0", "f1_1", "f2_1", "s1_0")

if __name__ == "__main__":
    for name in cases.CASES:
        out = cases.run_chain(O, name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: out[k] for k in KEEP})
        print(name, {k: float(np.nanmean(out[k])) for k in KEEP})
