"""Writes tests/golden/fftw_single_dct.npz: FFTW single-precision REDFT10 / REDFT01 outputs.

    python tests/golden/make_fftw_pin.py

Provenance: the vectors are copied, value for value, from
`scipy/fftpack/tests/fftw_single_ref.npz` (scipy 1.15.3 as installed in this image), which
scipy's maintainers generated with the real FFTW library in single precision
(`fftwf_plan_r2r_1d`, kinds REDFT10 = `dct_2_n` and REDFT01 = `dct_3_n`, input x = 0..n-1,
unnormalised; scipy/fftpack/tests/test_real_transforms.py::fftw_dct_ref reads them the same way).
They are the only FFTW-computed numbers available here: the reference's filter needs FFTW3
(src/nlkalman.c:6, 204-220, 278, 355), which this image does not have.

What they pin: FFTW's 1-D transform values at n = 4, 8, 12, 16 for one input. Together with the
reference's own scaling (src/nlkalman.c:281-298, 335-353), replayed in float by
tests/test_oracle.py, that fixes what `dct_threads_forward/inverse` return for a separable
input up to FFTW's 2-D codelet rounding. What they do NOT pin: the reference BINARY (no build
of it exists here), nor FFTW's rounding on other inputs.
"""
import os

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    src = os.path.join(os.path.dirname(scipy.__file__), "fftpack", "tests", "fftw_single_ref.npz")
    out = {}
    with np.load(src) as d:
        for n in (4, 8, 12, 16):
            out[f"redft10_{n}"] = d[f"dct_2_{n}"].astype(np.float32)
            out[f"redft01_{n}"] = d[f"dct_3_{n}"].astype(np.float32)
    np.savez(os.path.join(HERE, "fftw_single_dct.npz"), **out)
    print({k: v[:3] for k, v in out.items()})
