"""Full-size parity of the paths that round 2 checked at small sizes only (missing 3):
the spatial first-frame call (441-candidate windows in seven-round 4 x 2 blocks, reach-3 mask replay
over 269 grid rows), the second iteration of a first frame, 12 x 12 patches with a reach-2 replay,
and the deterministic aggregation mode against the serial oracle at 1080p.

Reference: src/nlkalman.c:597-600 (mask skip), :630-639 (spatial window), :930-931 (marking).
Tolerances as in test_gpu_parity.py: integer records exact, pixels max-abs 2e-3 / RMSE 2e-4 on the
0..255 scale. The only excused samples are pixels whose summed weight lies within 1e-4 relative of
the reference's absolute `aggr > 1e-6` threshold (src/nlkalman.c:939-942), where the order of a
float sum decides the side; they are counted and bounded."""
import numpy as np
import pytest

import cases
from test_gpu_parity import _check_records, _dev_frame, _to_o

pytestmark = pytest.mark.gpu


def _excuse_threshold_pixels(g, r, tr, what, most):
    edge = np.abs(tr["aggr"] - 1e-6) <= 1e-10
    assert int(edge.sum()) <= most, f"{what}: {int(edge.sum())} pixels sit at the aggregation threshold"
    return np.where(edge[..., None], r, g), int(edge.sum())


def test_spatial_first_frame_full_size_1080p(ctx, built, O, synth):
    """FLT1 spatial (deno0 = NULL) at 1920x1080 RGB sigma 20: every target searches the 21 x 21
    window (k = 50, groups of 20 reach 3 grid cells), then FLT2 spatial on the oracle's basic
    estimate (the first stage pair of BASELINE.json configs[4]). Serial oracle."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, _, c0 = synth.noisy_pair(w, h, ch, sigma, 1)
    o0 = built.rgb2opp(n0)
    p1, p2 = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.FLT2)
    r, tr = O.filter_frame(o0, None, None, sigma, _to_o(O, p1), trace=True)
    g, rec = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    assert 0.2 < 1 - tr["active"].mean() < 0.6          # the skip is exercised over the whole grid
    _check_records(rec, tr, "flt1 spatial 1080p")
    g, _ = _excuse_threshold_pixels(g, r, tr, "flt1 spatial 1080p", 64)
    cases.assert_close(g, r, "flt1 spatial 1080p")
    r2, tr2 = O.filter_frame(o0, None, r, sigma, _to_o(O, p2), trace=True)
    g2, rec2 = _dev_frame(ctx, False, o0, None, r, sigma, p2)
    _check_records(rec2, tr2, "flt2 spatial 1080p")
    g2, _ = _excuse_threshold_pixels(g2, r2, tr2, "flt2 spatial 1080p", 64)
    cases.assert_close(g2, r2, "flt2 spatial 1080p")
    clean = synth.clean_frame(w, h, ch, 0)
    assert abs(synth.psnr(built.opp2rgb(g2), clean) - synth.psnr(O.opp2rgb(r2), clean)) <= 0.02


def test_spatial_patch12_reach2_720p(ctx, built, O, synth):
    """FLT1 spatial with 12 x 12 patches (step 6, radius 10: a group reaches 2 grid cells, the
    second-order replay) at 1280x720 RGB sigma 40, serial oracle: `k_bm_topk<12,3,7>` blocks,
    `k_mask_commit_wave<2>`, `k_groupp<12>` with the wide halo."""
    w, h, ch, sigma = 1280, 720, 3, 40.0
    n0, _, _ = synth.noisy_pair(w, h, ch, sigma, 5)
    o0 = built.rgb2opp(n0)
    p = built.default_params(sigma, built.FLT1, patch_sz=12)
    r, tr = O.filter_frame(o0, None, None, sigma, _to_o(O, p), trace=True)
    g, rec = _dev_frame(ctx, False, o0, None, None, sigma, p)
    assert 0.02 < 1 - tr["active"].mean() < 0.6
    _check_records(rec, tr, "psz 12 spatial 720p")
    g, _ = _excuse_threshold_pixels(g, r, tr, "psz 12 spatial 720p", 64)
    cases.assert_close(g, r, "psz 12 spatial 720p")


def test_deterministic_mode_against_serial_oracle_1080p(built, O, synth):
    """nlk_ctx_set_deterministic at BASELINE.json configs[1]: FLT1 temporal at 1080p against the
    serial oracle with NO flip allowance beyond the threshold pixels (fixed summation order), and
    a second run bit-identical."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    det = built.Context(0)
    det.set_deterministic(True)
    try:
        prev, _ = _dev_frame(det, False, o0, None, None, sigma, p)
        g, rec = _dev_frame(det, False, o1, prev, None, sigma, p)
        g_again, _ = _dev_frame(det, False, o1, prev, None, sigma, p)
    finally:
        det.close()
    assert np.array_equal(g, g_again)
    r, tr = O.filter_frame(o1, prev, None, sigma, _to_o(O, p), trace=True)
    _check_records(rec, tr, "deterministic 1080p")
    g, n_edge = _excuse_threshold_pixels(g, r, tr, "deterministic 1080p", 64)
    cases.assert_close(g, r, "deterministic 1080p", flips=0)


@pytest.mark.parametrize("tail,single", [("0", "0"), ("7", "3"), ("1", "33"), ("60", "0")])
def test_group_tiles_of_mixed_sizes_1080p(ctx, built, O, synth, monkeypatch, tail, single):
    """`k_group8m` ends a 1080p launch on smaller tiles (3 x 1 targets, then single targets: tu_group8.hip).
    Whatever the split of the grid rows between the three kinds - none, odd counts, clamped counts - every target
    is filtered exactly once: FLT1 temporal and SMO1 equal the default split up to the order of the accumulator's
    adds, and the default equals the serial oracle."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, ps = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.SMO1)
    d0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    d1, rec = _dev_frame(ctx, False, o1, d0, None, sigma, p1)
    ds, _ = _dev_frame(ctx, True, d0, d1, None, sigma, ps)
    if tail == "0":   # (once: the default split against the oracle)
        r, tr = O.filter_frame(o1, d0, None, sigma, _to_o(O, p1), trace=True)
        _check_records(rec, tr, "flt1 temporal 1080p")
        g, _ = _excuse_threshold_pixels(d1, r, tr, "flt1 temporal 1080p", 64)
        cases.assert_close(g, r, "flt1 temporal 1080p")
    monkeypatch.setenv("NLK_G8_TAIL", tail)
    monkeypatch.setenv("NLK_G8_SINGLE", single)
    e1, _ = _dev_frame(ctx, False, o1, d0, None, sigma, p1)
    es, _ = _dev_frame(ctx, True, d0, d1, None, sigma, ps)
    cases.assert_close(e1, d1, f"tail {tail} single {single}: flt1 temporal", flips=40)
    cases.assert_close(es, ds, f"tail {tail} single {single}: smo1", flips=40)


def test_single_target_tail_on_single_row_tiles_720p(ctx, built, O, synth, monkeypatch):
    """Grids that use tiles of one grid row (2 x 1 targets: 720p, and every first frame) end their launch on single
    targets too. FLT1 spatial + temporal at 1280x720 against the serial oracle, and against the same calls with the
    tail switched off / enlarged."""
    w, h, ch, sigma = 1280, 720, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 9)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1)
    d0, rec0 = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    d1, rec1 = _dev_frame(ctx, False, o1, d0, None, sigma, p1)
    r0, tr0 = O.filter_frame(o0, None, None, sigma, _to_o(O, p1), trace=True)
    _check_records(rec0, tr0, "720p spatial")
    g0, _ = _excuse_threshold_pixels(d0, r0, tr0, "720p spatial", 64)
    cases.assert_close(g0, r0, "720p spatial")
    r1, tr1 = O.filter_frame(o1, d0, None, sigma, _to_o(O, p1), trace=True)
    _check_records(rec1, tr1, "720p temporal")
    g1, _ = _excuse_threshold_pixels(d1, r1, tr1, "720p temporal", 64)
    cases.assert_close(g1, r1, "720p temporal")
    for single in ("0", "5", "40"):
        monkeypatch.setenv("NLK_G8_SINGLE", single)
        e0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        e1, _ = _dev_frame(ctx, False, o1, d0, None, sigma, p1)
        cases.assert_close(e0, d0, f"single {single}: spatial", flips=40)
        cases.assert_close(e1, d1, f"single {single}: temporal", flips=40)


def test_4k_patch8_temporal_against_serial_oracle(ctx, built, O, synth):
    """3840x2160 RGB sigma 20 with the default 8x8 patches, FLT1 temporal, serial oracle: the patch grid of
    959 x 539 targets (4x the 1080p one) through the 8-wavefront match tiles, the row replay of the mask and
    `k_group8m` with its three kinds of tiles (3x2, the last 8 grid rows 3x1, the last 3 one target each).
    Records exact, pixels within the tolerances of the 1080p tests."""
    w, h, ch, sigma = 3840, 2160, 3, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 3)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p)
    g, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    r, tr = O.filter_frame(o1, prev, None, sigma, _to_o(O, p), trace=True)
    _check_records(rec, tr, "4K psz 8")
    g, _ = _excuse_threshold_pixels(g, r, tr, "4K psz 8", 256)
    cases.assert_close(g, r, "4K psz 8", flips=160)
    assert abs(synth.psnr(built.opp2rgb(g), c1) - synth.psnr(O.opp2rgb(r), c1)) <= 0.02


def test_host_pointer_calls_pipeline_the_frame_in_row_bands(ctx, built, synth, monkeypatch):
    """The drop-in API (host pointers, libnlkalman.so -> nlk_filter_frame_host) moves a frame over PCIe in
    row bands while the bands before are matched and filtered, and returns finished rows while the last
    bands are filtered. Same decisions as the whole-frame device call, so: equal up to the order of the
    accumulator's atomic adds, for every kind of call of the flt1 -> flt2 -> smo1 chain at 1080p (incl. a
    previous frame with a NaN ring and an occlusion hole: validity map built band by band), and for 2 and
    8 bands."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, p2, ps = (built.default_params(sigma, m) for m in (built.FLT1, built.FLT2, built.SMO1))
    d0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    hole = d0.copy()
    hole[:1], hole[-2:], hole[:, :1], hole[:, -2:] = np.nan, np.nan, np.nan, np.nan   # the warp's NaN ring
    hole[500:540, 900:1000] = np.nan
    d1, _ = _dev_frame(ctx, False, o1, hole, None, sigma, p1)
    d2, _ = _dev_frame(ctx, False, o1, hole, d1, sigma, p2)
    ds, _ = _dev_frame(ctx, True, d0, d2, None, sigma, ps)
    for bands in (None, "2", "8"):
        if bands:
            monkeypatch.setenv("NLK_HOST_BANDS", bands)
        tag = f"host bands {bands or 'default'}"
        cases.assert_close(built.filter_frame(o0, None, None, sigma, p1), d0, f"{tag}: flt1 spatial", flips=40)
        cases.assert_close(built.filter_frame(o1, hole, None, sigma, p1), d1, f"{tag}: flt1 temporal", flips=40)
        if bands is None:
            cases.assert_close(built.filter_frame(o1, hole, d1, sigma, p2), d2, f"{tag}: flt2", flips=40)
            cases.assert_close(built.smooth_frame(d0, d2, None, sigma, ps), ds, f"{tag}: smo1", flips=40)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_host_pointer_pipeline_on_odd_sizes(ctx, built, synth, seed):
    """Row bands of the host-pointer calls on frames whose sizes are nothing round: one or three channels,
    8 x 8 and 12 x 12 patches (mask reach 1 and 0), a previous frame with NaN holes on band seams, a basic
    estimate. Equal to the device call up to the order of the accumulator's atomic adds."""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(500, 900)), int(rng.integers(420, 700))
    ch = int(rng.choice([1, 3]))
    psz = int(rng.choice([8, 12]))
    sigma = float(rng.choice([20.0, 40.0]))
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, seed)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1, patch_sz=psz)
    p2 = built.default_params(sigma, built.FLT2, patch_sz=psz)
    d0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    hole = d0.copy()
    for _ in range(12):   # (holes anywhere: some straddle the seams between two bands)
        y, x = int(rng.integers(0, h - 20)), int(rng.integers(0, w - 30))
        hole[y:y + int(rng.integers(1, 20)), x:x + int(rng.integers(1, 30))] = np.nan
    d1, _ = _dev_frame(ctx, False, o1, hole, None, sigma, p1)
    d2, _ = _dev_frame(ctx, False, o1, hole, d1, sigma, p2)
    what = f"host pipeline {w}x{h}x{ch} psz {psz}"
    cases.assert_close(built.filter_frame(o0, None, None, sigma, p1), d0, what + ": flt1 spatial", flips=40)
    cases.assert_close(built.filter_frame(o1, hole, None, sigma, p1), d1, what + ": flt1 temporal", flips=40)
    cases.assert_close(built.filter_frame(o1, hole, d1, sigma, p2), d2, what + ": flt2", flips=40)
