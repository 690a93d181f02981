"""Seeded parity cases shared by the golden-fixture generator, the CPU tests of
the oracle and the GPU parity tests. Inputs are rebuilt from seeds (synthetic
pattern + the reference tool chain's LCG AWGN); only expected OUTPUTS are
stored under tests/golden/ (made by tests/golden/make_golden.py)."""
import importlib

import numpy as np

synth = importlib.import_module("bwd-nlkalman_amd.synth")

FLT1, FLT2, SMO1 = 0, 1, 2

# name: (w, h, ch, sigma, overrides)
CASES = {
    "gray64_s20": (64, 64, 1, 20.0, {}),
    "rgb96x64_s20": (96, 64, 3, 20.0, {}),
    "rgb72x48_s40": (72, 48, 3, 40.0, {}),
    "rgb84x60_p12_s40": (84, 60, 3, 40.0, {"patch_sz": 12}),
    "gray70x53_ragged": (70, 53, 1, 20.0, {}),       # (w-psz) % step != 0
    "rgb40x40_p4": (40, 40, 3, 20.0, {"patch_sz": 4, "search_sz_x": 6}),
}


def flow_and_occ(w, h):
    flow = np.zeros((h, w, 2), np.float32)
    flow[..., 0] = 2.0 + 0.3 * np.sin(np.arange(w) / 9.0)[None, :]
    flow[..., 1] = 0.25 + 0.2 * np.cos(np.arange(h) / 7.0)[:, None]
    occ = np.zeros((h, w), np.float32)
    occ[h // 3:h // 3 + 10, w // 3:w // 3 + 15] = 255
    occ[5, 7] = 255
    return flow, occ


def inputs(name):
    w, h, ch, sigma, over = CASES[name]
    seed = sum(map(ord, name))
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, seed)
    flow, occ = flow_and_occ(w, h)
    return dict(w=w, h=h, ch=ch, sigma=sigma, over=over, n0=n0, n1=n1, clean1=c1,
                flow=flow, occ=occ)


def run_chain(B, name):
    """The per-frame pipeline of scripts/nlkalman-seq.sh on two frames, through
    backend B (the oracle module or the product package — same function names):
    frame 0 spatial FLT1 -> FLT2; frame 1 warp + temporal FLT1 -> FLT2; then
    SMO1 of frame 0 against frame 1's result. Returns every stage output."""
    I = inputs(name)
    s, over = I["sigma"], I["over"]
    p1 = B.default_params(s, FLT1, **over)
    p2 = B.default_params(s, FLT2, **over)
    ps = B.default_params(s, SMO1, **{k: v for k, v in over.items() if k != "search_sz_x"})
    o0, o1 = B.rgb2opp(I["n0"]), B.rgb2opp(I["n1"])
    out = {}
    out["f1_0"] = B.filter_frame(o0, None, None, s, p1)
    out["f2_0"] = B.filter_frame(o0, None, out["f1_0"], s, p2)
    out["w1"] = B.warp_bicubic(out["f1_0"], I["flow"], I["occ"])
    out["w2"] = B.warp_bicubic(out["f2_0"], I["flow"], I["occ"])
    out["f1_1"] = B.filter_frame(o1, out["w1"], None, s, p1)
    out["f2_1"] = B.filter_frame(o1, out["w2"], out["f1_1"], s, p2)
    out["ws"] = B.warp_bicubic(out["f2_1"], -I["flow"], I["occ"])
    out["s1_0"] = B.smooth_frame(out["f2_0"], out["ws"], None, s, ps)
    out["rgb_f2_1"] = B.opp2rgb(out["f2_1"])
    return out


def run_chain_stagewise(B, ref, name):
    """Same pipeline, but every stage of backend B is fed the REFERENCE outputs
    `ref` of the previous stages, so that each stage is compared in isolation
    (no error accumulation, no mask-order divergence carried forward)."""
    I = inputs(name)
    s, over = I["sigma"], I["over"]
    p1 = B.default_params(s, FLT1, **over)
    p2 = B.default_params(s, FLT2, **over)
    ps = B.default_params(s, SMO1, **{k: v for k, v in over.items() if k != "search_sz_x"})
    o0, o1 = B.rgb2opp(I["n0"]), B.rgb2opp(I["n1"])
    out = {}
    out["f1_0"] = B.filter_frame(o0, None, None, s, p1)
    out["f2_0"] = B.filter_frame(o0, None, ref["f1_0"], s, p2)
    out["w1"] = B.warp_bicubic(ref["f1_0"], I["flow"], I["occ"])
    out["w2"] = B.warp_bicubic(ref["f2_0"], I["flow"], I["occ"])
    out["f1_1"] = B.filter_frame(o1, ref["w1"], None, s, p1)
    out["f2_1"] = B.filter_frame(o1, ref["w2"], ref["f1_1"], s, p2)
    out["ws"] = B.warp_bicubic(ref["f2_1"], -I["flow"], I["occ"])
    out["s1_0"] = B.smooth_frame(ref["f2_0"], ref["ws"], None, s, ps)
    out["rgb_f2_1"] = B.opp2rgb(ref["f2_1"])
    return out


def assert_close(got, ref, what, maxabs=2e-3, rmse=2e-4, flips=0):
    """Float tolerance of the parity bar (0..255 scale). The FP noise floor of
    the path is 2e-4 max-abs / 2.5e-5 RMSE; `flips` = number of
    samples allowed to exceed maxabs (pixels whose aggregation weight sits at
    the reference's absolute 1e-6 threshold flip between denoised and noisy)."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, what
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), f"{what}: NaN pattern differs"
    d = np.abs(np.where(nan_r, 0, got - ref))
    nbad = int((d > maxabs).sum())
    assert nbad <= flips, f"{what}: {nbad} samples differ by > {maxabs} (max {d.max():.3e})"
    r = float(np.sqrt(np.mean(np.minimum(d, maxabs) ** 2)))
    assert r <= rmse, f"{what}: rmse {r:.3e} > {rmse}"


def excuse_threshold_pixels(g, r, tr, what, most):
    """The only samples a comparison with the oracle may excuse: pixels whose summed weight lies within
    1e-4 relative of the reference's absolute `aggr > 1e-6` threshold (src/nlkalman.c:939-942), where the
    order of a float sum decides whether the pixel is normalised or passed through. Counted and bounded;
    returns g with those pixels taken from r, and their number."""
    edge = np.abs(tr["aggr"] - 1e-6) <= 1e-10
    n = int(edge.sum())
    assert n <= most, f"{what}: {n} pixels sit at the aggregation threshold (at most {most} expected)"
    return np.where(edge[..., None], r, g), n


def excuse_flips(a, b, cur, what, most, tol=5e-4):
    """Two runs of the PRODUCT on the same inputs (different tilings / bands / strips) add the accumulator in
    different orders, so a pixel whose summed weight sits at the 1e-6 threshold may be normalised in one and
    passed through in the other. Such a pixel carries its signature: in exactly one of the two outputs it
    equals the input `cur` bit for bit, in every channel, and the two outputs differ by more than `tol` there
    (a pixel that a smoother's pass-through patch dominates equals the input up to rounding in both). Those
    pixels - and no others - are excused, counted and bounded; returns a with them taken from b, and their
    number."""
    a, b, cur = np.asarray(a), np.asarray(b), np.asarray(cur)
    pa, pb = (a == cur).all(-1), (b == cur).all(-1)
    with np.errstate(invalid="ignore"):
        flip = (pa != pb) & (np.nan_to_num(np.abs(a - b)).max(-1) > tol)
    n = int(flip.sum())
    assert n <= most, f"{what}: {n} pixels flipped at the aggregation threshold (at most {most} expected)"
    return np.where(flip[..., None], b, a), n


def blob_mask(w, h, seed, nblobs=9):
    """A blob-shaped occlusion mask (0 / 255, any non-zero = occluded: src/nlkalman.c:77): a few filled
    ellipses of different sizes, some cut by the image border."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), np.float32)
    for _ in range(nblobs):
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        ry, rx = rng.integers(3, max(4, h // 12)), rng.integers(3, max(4, w // 12))
        m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
    m[5, 7] = 255   # (a single occluded pixel)
    return m
