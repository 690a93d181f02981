"""Generates tests/golden/tvl1_ref_64x48.npz with the REFERENCE's TV-L1 build
(oracle/_ref/libtvl1flow_ref.so = lib/tvl1flow compiled from /root/reference by
`make -C oracle ref`). Run in the build container:  python tests/golden/make_tvl1_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O            # noqa: E402  (only for the scale count rule of main.c)
from test_tvl1 import _pair   # noqa: E402

L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtvl1flow_ref.so"))
C.CDLL("libgomp.so.1").omp_set_num_threads(1)
fp, i, f = C.POINTER(C.c_float), C.c_int, C.c_float
L.Dual_TVL1_optic_flow_multiscale.argtypes = [fp, fp, fp, fp, i, i, f, f, f, i, i, f, i, f, C.c_bool]
w, h, lam = 64, 48, 0.4
i0, i1 = _pair(w, h, 11)
u, v = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
ns = O.tvl1_auto_scales(w, h)
L.Dual_TVL1_optic_flow_multiscale(i0.ctypes.data_as(fp), i1.ctypes.data_as(fp), u.ctypes.data_as(fp),
                                  v.ctypes.data_as(fp), w, h, 0.25, lam, 0.3, ns, 0, 0.5, 5, 0.01, False)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tvl1_ref_64x48.npz"), i0=i0, i1=i1, u=u, v=v,
                    lam=np.float32(lam))
print("scales", ns, "median flow", np.median(u), np.median(v))
