"""In-process sequence driver (bwd-nlkalman_amd/sequence.py): the recursion of
scripts/nlkalman-seq.sh with every frame resident on the GPU, against the same recursion written
with the CPU oracle's functions (flow, mask, warp, filters, smoother)."""
import importlib

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
W, H, CH, SIGMA, NF = 96, 64, 3, 20.0, 3


def _frames(synth):
    return [synth.awgn(synth.clean_frame(W, H, CH, t), SIGMA, 100 + t) for t in range(NF)]


def _lum(rgb):
    c = rgb.astype(np.float64)
    return (.299 * c[..., 0] + .587 * c[..., 1] + .114 * c[..., 2]).astype(np.float32)


def _oracle_chain(O, frames, lam, fscale, th):
    p1, p2, ps = (O.default_params(SIGMA, m) for m in (O.FLT1, O.FLT2, O.SMO1))
    f1s, f2s = [], []
    for t, rgb in enumerate(frames):
        nz = O.rgb2opp(rgb)
        if t == 0:
            f1 = O.filter_frame(nz, None, None, SIGMA, p1)
            f2 = O.filter_frame(nz, None, f1, SIGMA, p2)
        else:
            u, v = O.tvl1_flow(_lum(rgb), _lum(O.opp2rgb(f2s[-1])), lam=lam, fscale=fscale)
            fl = np.stack([u, v], -1)
            occ = O.tvl1_occlusion_mask(fl, th)
            f1 = O.filter_frame(nz, O.warp_bicubic(f1s[-1], fl, occ), None, SIGMA, p1)
            f2 = O.filter_frame(nz, O.warp_bicubic(f2s[-1], fl, occ), f1, SIGMA, p2)
        f1s.append(f1)
        f2s.append(f2)
    smo = [None] * NF
    smo[-1] = f2s[-1]
    for t in range(NF - 2, -1, -1):
        u, v = O.tvl1_flow(_lum(O.opp2rgb(f2s[t])), _lum(O.opp2rgb(smo[t + 1])), lam=lam, fscale=fscale)
        fl = np.stack([u, v], -1)
        occ = O.tvl1_occlusion_mask(fl, th)
        smo[t] = O.smooth_frame(f2s[t], O.warp_bicubic(smo[t + 1], fl, occ), None, SIGMA, ps)
    return f1s, f2s, smo


def test_resident_sequence_equals_oracle_recursion(ctx, built, O, synth):
    """Stage by stage: every step of the driver is checked against the oracle's functions fed
    with the DRIVER's own previous outputs (the recursion amplifies 1e-4 differences through
    the iterative flow and the thresholded occlusion mask, so a free-running oracle chain can
    only be compared in PSNR: done at the end)."""
    seq = importlib.import_module("bwd-nlkalman_amd.sequence")
    frames = _frames(synth)
    lam, fscale, th = 0.40, 0, 0.75
    p1, p2, ps = (O.default_params(SIGMA, m) for m in (O.FLT1, O.FLT2, O.SMO1))
    sf = seq.SequenceFilter(ctx, W, H, CH, SIGMA, of_lambda=lam, of_fscale=fscale, occ_th=th)
    got1, got2 = [], []
    for t, rgb in enumerate(frames):
        d = ctx.upload(rgb)
        sf.push(d)
        ctx.free(d)
        got1.append(ctx.download(sf.flt1, (H, W, CH)))
        got2.append(ctx.download(sf.flt2, (H, W, CH)))
        nz = O.rgb2opp(rgb)
        if t == 0:
            w1 = w2 = None
        else:
            u, v = O.tvl1_flow(_lum(rgb), _lum(O.opp2rgb(got2[t - 1])), lam=lam, fscale=fscale)
            fl = np.stack([u, v], -1)
            occ = O.tvl1_occlusion_mask(fl, th)
            w1, w2 = O.warp_bicubic(got1[t - 1], fl, occ), O.warp_bicubic(got2[t - 1], fl, occ)
        for name, got, (r, tr) in (("flt1", got1[t], O.filter_frame(nz, w1, None, SIGMA, p1, trace=True)),
                                   ("flt2", got2[t], O.filter_frame(nz, w2, got1[t], SIGMA, p2, trace=True))):
            g, _ = cases.excuse_threshold_pixels(got, r, tr, f"{name} frame {t}", 16)
            cases.assert_close(g, r, f"{name} frame {t}")
    smo = [ctx.download(d, (H, W, CH)) for d in sf.smooth()]
    assert np.array_equal(smo[-1], got2[-1])
    for t in range(NF - 2, -1, -1):
        u, v = O.tvl1_flow(_lum(O.opp2rgb(got2[t])), _lum(O.opp2rgb(smo[t + 1])), lam=lam, fscale=fscale)
        fl = np.stack([u, v], -1)
        occ = O.tvl1_occlusion_mask(fl, th)
        ref, tr = O.smooth_frame(got2[t], O.warp_bicubic(smo[t + 1], fl, occ), None, SIGMA, ps, trace=True)
        g, _ = cases.excuse_threshold_pixels(smo[t], ref, tr, f"smo1 frame {t}", 16)
        cases.assert_close(g, ref, f"smo1 frame {t}")
    assert len(sf.flow_iterations) == 2 * (NF - 1) and min(sf.flow_iterations) > 0
    # free-running oracle chain: same quality
    r1, r2, rs = _oracle_chain(O, frames, lam, fscale, th)
    for t in range(NF):
        clean = O.rgb2opp(synth.clean_frame(W, H, CH, t))
        for name, g, r in (("flt1", got1, r1), ("flt2", got2, r2), ("smo1", smo, rs)):
            assert abs(synth.psnr(g[t], clean) - synth.psnr(r[t], clean)) < 0.02, f"{name} frame {t}"
    clean = O.rgb2opp(synth.clean_frame(W, H, CH, NF - 1))
    noisy = O.rgb2opp(frames[-1])
    assert synth.psnr(got2[-1], clean) > synth.psnr(got1[-1], clean) - 0.3 > synth.psnr(noisy, clean) + 3
    rgb = sf.download_rgb(sf.flt2)
    assert np.abs(rgb - O.opp2rgb(got2[-1])).max() < 1e-4


def test_sequence_without_history_frees_frames(ctx, built, synth):
    seq = importlib.import_module("bwd-nlkalman_amd.sequence")
    sf = seq.SequenceFilter(ctx, W, H, 1, SIGMA, keep_history=False)
    for t in range(2):
        d = ctx.upload(np.ascontiguousarray(_frames(synth)[t][..., :1]))
        sf.push(d)
        ctx.free(d)
    assert np.isfinite(ctx.download(sf.flt2, (H, W, 1))).all()
    with pytest.raises(RuntimeError):
        sf.smooth()


def test_one_process_tool_equals_the_driver(ctx, built, O, synth, tmp_path):
    """bin/nlkalman-seq = scripts/nlkalman-seq.sh in one process: same positional arguments, same
    files in the output folder; its frames equal the Python driver's (same C-ABI calls)."""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_cli import run
    seq = importlib.import_module("bwd-nlkalman_amd.sequence")
    frames = _frames(synth)
    src = tmp_path / "in"
    src.mkdir()
    # float TIFF inputs: PFM written here (top row first, no flip), converted by nlk-imgconv
    from test_cli import rpfm, wpfm
    for t, f in enumerate(frames):
        wpfm(src / f"{t + 3:03d}.pfm", f)
        r = run("nlk-imgconv", src / f"{t + 3:03d}.pfm", src / f"{t + 3:03d}.tif")
        assert r.returncode == 0, r.stderr
    out = tmp_path / "out"
    # (both sides aggregate deterministically - NLK_DETERMINISTIC / nlk_ctx_set_deterministic -: two free-running
    # recursions whose float atomics differ in the last bits are amplified by the iterative flow and the thresholded
    # occlusion mask at a few pixels, and a statistical comparison of them failed once in a dozen runs)
    r = run("nlkalman-seq", src / "%03d.tif", 3, 3 + NF - 1, SIGMA, out, 1, "", "", "0 0.40 0.75 0 0.40 0.75",
            env=dict(os.environ, NLK_DETERMINISTIC="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    names = sorted(os.listdir(out))
    for t in range(3, 3 + NF):
        assert f"flt1-{t:03d}.tif" in names and f"flt2-{t:03d}.tif" in names and f"smo1-{t:03d}.tif" in names
    assert "bflo1-004.flo" in names and "bocc1-004.png" in names and "fflo-003.flo" in names and "focc-003.png" in names
    assert "bflo1-003.flo" not in names

    def rd(name):
        r2 = run("nlk-imgconv", out / name, tmp_path / "x.pfm")
        assert r2.returncode == 0, r2.stderr
        return rpfm(tmp_path / "x.pfm")
    ctx.set_deterministic(True)
    try:
        sf = seq.SequenceFilter(ctx, W, H, CH, SIGMA, of_lambda=0.40, of_fscale=0, occ_th=0.75)
        for f in frames:
            d = ctx.upload(f)
            sf.push(d)
            ctx.free(d)
        smo = sf.smooth()
        want2 = sf.download_rgb(sf.flt2)
        want_s = sf.download_rgb(smo[0])
    finally:
        ctx.set_deterministic(False)
    # the same C-ABI calls in the same order with bit-reproducible aggregation: the same frames (the tool's frames
    # went through 32-bit float TIFF files, which is exact)
    for got, want, what in ((rd(f"flt2-{3 + NF - 1:03d}.tif"), want2, "flt2 of the last frame"),
                            (rd("smo1-003.tif"), want_s, "smo1 of the first frame")):
        d = np.abs(got - want)
        assert np.quantile(d, 0.99) < 2e-3 and np.sqrt(np.mean(d ** 2)) < 5e-2, what
        assert d.max() < 1e-3, (what, float(d.max()))
    # usage / error paths need no GPU work
    assert run("nlkalman-seq").returncode == 1
    r = run("nlkalman-seq", src / "%03d.tif", 3, 9, SIGMA, out)
    assert r.returncode == 1 and "not found" in r.stdout
