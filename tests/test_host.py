"""CPU tests of the boundary: the C-ABI libraries load without a GPU, export every
symbol the headers declare, keep the reference's struct layout, and fail loudly
(never fall back to a CPU path) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _declared(header):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(\w+)\s*\([^;{]*\)\s*;", txt)))


def test_headers_declare_expected_symbols(built):
    api = _declared("nlkalman.h")
    assert api == sorted(built.API_SYMBOLS)
    abi = _declared("nlk_hip.h")
    assert abi == sorted(built.HIP_SYMBOLS)
    assert _declared("tvl1flow.h") == sorted(built.TVL1_SYMBOLS)


def test_libraries_export_every_declared_symbol(built):
    hip, api = built.hip(), built.api()
    for s in _declared("nlk_hip.h"):
        assert hasattr(hip, s), s
    for s in _declared("nlkalman.h") + _declared("tvl1flow.h"):
        assert hasattr(api, s), s


def test_struct_layout_matches_reference(built):
    """reference: src/nlkalman.h:22-37 — 6 int + 3 float = 36 bytes, by value."""
    assert C.sizeof(built.Params) == 36
    names = [n for n, _ in built.Params._fields_]
    assert names == ["patch_sz", "search_sz_x", "search_sz_t", "npatches_x", "npatches_t",
                     "npatches_tagg", "dista_lambda", "beta_x", "beta_t"]
    assert (built.FLT1, built.FLT2, built.SMO1) == (0, 1, 2)


def test_default_params_match_oracle(built, O):
    for sigma in (2.0, 10.0, 20.0, 40.0, 63.5):
        for mode in (0, 1, 2):
            assert built.default_params(sigma, mode).as_dict() == O.default_params(sigma, mode).as_dict()
    p = built.default_params(20.0, 0, patch_sz=12, beta_x=1.5)
    assert p.patch_sz == 12 and p.beta_x == 1.5 and p.npatches_t == 30


def test_no_gpu_fails_loudly(built):
    if built.hip().nlk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(built.NlkError, match="no HIP device"):
        built.Context(0)
    # the void-returning drop-in API exits(1) with a message, like the reference's fatal paths
    code = ("import importlib,numpy as np,sys; sys.path.insert(0, %r);"
            "p=importlib.import_module('bwd-nlkalman_amd');"
            "p.filter_frame(np.zeros((16,16,1),np.float32),None,None,20,p.default_params(20,0))" % ROOT)
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot initialise the GPU" in r.stderr


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/
    may name it."""
    bad = []
    for base in ("bwd-nlkalman_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hip", ".cpp")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"import\s+oracle|from\s+oracle|nlk_oracle|nlko_|oracle/|libnlk_oracle", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
