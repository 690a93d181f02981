"""Supplementary evidence, NOT a pin (see oracle/nlk_oracle.c header).

During the survey of the reference a build of its own sources (src/nlkalman.c,
main-flt.c, main-smo.c + lib/iio, lib/argparse) was left in /tmp/oracle of the
build container, linked against a table-based DCT stand-in because FFTW3 is
not installed. Such a build is not a legitimate reference build (the DCT is a
stand-in), so the oracle stays "parity unpinned"; but everything except the
FFTW rounding — block matching, qsort order, Welford statistics, gains,
aggregation, mask skip, warp, colour transform, CLI plumbing — is the
reference's own compiled code. This script runs those binaries (serial builds:
nlkalman-flt-noomp, nlkalman-smo-serial) on seeded synthetic inputs and stores
their outputs under tests/golden/survey_shim_*.npz; tests/test_oracle.py
compares the oracle with them (max-abs <= 1e-3 on the 0..255 scale).

Only runnable in the container that still holds /tmp/oracle.
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

BIN = "/tmp/oracle"


def wpfm(path, a):
    a = np.ascontiguousarray(a, np.float32)
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    with open(path, "wb") as f:
        f.write(b"%s\n%d %d\n-1.0\n" % (b"PF" if ch == 3 else b"Pf", w, h))
        f.write(a.tobytes())


def rpfm(path):
    with open(path, "rb") as f:
        t = f.readline().strip()
        w, h = map(int, f.readline().split())
        f.readline()
        ch = 3 if t == b"PF" else 1
        return np.frombuffer(f.read(), np.float32).reshape(h, w, ch).copy()


def wflo(path, fl):
    h, w = fl.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PIEH" + struct.pack("<ii", w, h) + np.ascontiguousarray(fl, np.float32).tobytes())


def main():
    for name in ("rgb72x48_s40", "gray64_s20"):
        I = cases.inputs(name)
        S = "%g" % I["sigma"]
        with tempfile.TemporaryDirectory() as d:
            p = lambda f: os.path.join(d, f)  # noqa: E731
            wpfm(p("n0.pfm"), I["n0"]); wpfm(p("n1.pfm"), I["n1"])
            wflo(p("b.flo"), I["flow"]); wflo(p("f.flo"), -I["flow"]); wpfm(p("occ.pfm"), I["occ"])
            flt = os.path.join(BIN, "nlkalman-flt-noomp")
            subprocess.check_call([flt, "-i", p("n0.pfm"), "-s", S, "--flt11", p("f1_0.pfm"), "--flt21", p("f2_0.pfm")])
            subprocess.check_call([flt, "-i", p("n1.pfm"), "-s", S, "-o", p("b.flo"), "-k", p("occ.pfm"),
                                   "--flt10", p("f1_0.pfm"), "--flt20", p("f2_0.pfm"),
                                   "--flt11", p("f1_1.pfm"), "--flt21", p("f2_1.pfm")])
            subprocess.call([os.path.join(BIN, "nlkalman-smo-serial"), "--flt1", p("f2_0.pfm"), "--smo0", p("f2_1.pfm"),
                             "-o", p("f.flo"), "-k", p("occ.pfm"), "--smo1", p("s1_0.pfm"), "-s", S])
            out = {k: rpfm(p(k + ".pfm")) for k in ("f1_0", "f2_0", "f1_1", "f2_1", "s1_0")}
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "survey_shim_" + name + ".npz"), **out)
        print(name, {k: float(v.mean()) for k, v in out.items()})


if __name__ == "__main__":
    main()
