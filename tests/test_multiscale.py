"""Multiscale wrapper (reference: lib/multiscale/): whole-image DCT on the
GPU, the three tools, against the CPU restatement (oracle/ms_oracle.c, parity unpinned: FFTW is
not available) and against scipy's DCTs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bwd-nlkalman_amd", "bin")


def _img(w, h, ch, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 120 + 70 * np.sin(0.11 * x + 0.07 * y) * np.cos(0.05 * x - 0.13 * y)
    return (base[..., None] + rng.normal(0, 12, (h, w, ch))).astype(np.float32)


def test_oracle_dct_is_fftw_redft_with_reference_scaling(O):
    """REDFT10 / (4 rows cols) and REDFT01 = scipy's unnormalised type-2 / type-3 transforms."""
    import scipy.fft as sf
    for (w, h, ch) in [(22, 13, 3), (16, 16, 1), (37, 50, 3)]:
        a = _img(w, h, ch, w + h)
        f = O.ms_dct(a)
        ref = sf.dctn(a.astype(np.float64), type=2, axes=(0, 1)) / (4 * w * h)
        assert np.abs(f - ref).max() < 1e-4 * np.abs(ref).max()
        back = O.ms_dct(f, inverse=True)
        assert np.abs(back - sf.dctn(f.astype(np.float64), type=3, axes=(0, 1))).max() < 1e-3
        assert np.abs(back - a).max() < 1e-3
        assert abs(f[0, 0].mean() - a.mean()) < 1e-3  # the DC coefficient is the mean: levels keep the range


def test_oracle_pyramid_round_trip(O):
    a = _img(96, 72, 3, 1)
    lv = O.ms_decompose(a, 3, 2.0)
    assert [x.shape for x in lv] == [(72, 96, 3), (36, 48, 3), (18, 24, 3)]
    assert np.abs(O.ms_recompose(lv, 0.8) - a).max() < 1e-3
    # a level whose low frequencies were changed shows up in the result, the rest does not
    lv2 = [x.copy() for x in lv]
    lv2[2] += 10.0
    r = O.ms_recompose(lv2, 0.8)
    assert abs((r - a).mean() - 10.0) < 1e-2


def test_tools_usage(built):
    if not os.path.exists(os.path.join(BIN, "decompose")):
        built.build()
    for tool, usage in (("decompose", "input prefix levels suffix [-r ratio]"),
                        ("recompose", "prefix levels suffix output [-c factor]"),
                        ("merge_coarse", "image coarse result [-c factor]")):
        r = subprocess.run([os.path.join(BIN, tool)], capture_output=True, text=True)
        assert r.returncode == 1 and r.stderr.strip() == f"Usage: {os.path.join(BIN, tool)} {usage}"
        r = subprocess.run([os.path.join(BIN, tool), "-h"], capture_output=True, text=True)
        assert r.returncode == 1 and "Usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ch", [(22, 13, 3), (64, 64, 1), (100, 75, 3), (333, 190, 3)])
def test_gpu_image_dct_vs_oracle(ctx, built, O, w, h, ch):
    a = _img(w, h, ch, w)
    d = ctx.upload(a)
    ctx.image_dct(d, w, h, ch, False)
    f = ctx.download(d, (h, w, ch))
    ref = O.ms_dct(a)
    assert np.abs(f - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    ctx.image_dct(d, w, h, ch, True)
    back = ctx.download(d, (h, w, ch))
    assert np.abs(back - a).max() < 2e-3           # f32 matrix products over up to 333 terms
    ctx.free(d)


@pytest.mark.gpu
def test_tools_pyramid_files(built, O, tmp_path):
    """decompose -> recompose / merge_coarse on files, the way scripts/msnlkalman-seq.sh:56-60,
    106-110 calls them, against the oracle's pyramid."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cli import rpfm, run, wpfm
    a = _img(120, 90, 3, 5)
    wpfm(tmp_path / "in.pfm", a)
    r = run("decompose", tmp_path / "in.pfm", str(tmp_path / "ms"), 3, "-007.pfm", "-r", 2)
    assert r.returncode == 0, r.stderr
    lv = [rpfm(tmp_path / f"ms{i}-007.pfm") for i in range(3)]
    want = O.ms_decompose(a, 3, 2.0)
    for g, wv in zip(lv, want):
        assert g.shape == wv.shape and np.abs(g - wv).max() < 2e-3
    r = run("recompose", str(tmp_path / "ms"), 3, "-007.pfm", tmp_path / "out.pfm", "-c", 0.8)
    assert r.returncode == 0, r.stderr
    out = rpfm(tmp_path / "out.pfm")
    assert np.abs(out - O.ms_recompose(want, 0.8)).max() < 3e-3 and np.abs(out - a).max() < 3e-3
    r = run("merge_coarse", tmp_path / "in.pfm", tmp_path / "ms1-007.pfm", tmp_path / "mc.pfm")
    assert r.returncode == 0, r.stderr
    assert np.abs(rpfm(tmp_path / "mc.pfm") - O.ms_recompose([a, want[1]], 0.8)).max() < 3e-3
