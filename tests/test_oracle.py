"""CPU tests of the oracle (oracle/nlk_oracle.c): known answers derivable from
the reference's source, the committed golden fixtures, an independent
numpy/scipy restatement, and the supplementary survey-build vectors."""
import os

import numpy as np
import pytest
from scipy.fft import dctn, idctn

import cases
import ref_numpy

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------ known answers

def test_default_params_table(O):
    """SURVEY.md Appendix C / reference: src/nlkalman.c:456-486."""
    want = {
        (20, O.FLT1): (50, 3.11, 30, 20, 1.95), (20, O.FLT2): (20, 0.29, 20, 1, 1.66),
        (20, O.SMO1): (0, 0.0, 45, 45, 5.2), (40, O.FLT1): (60, 2.31, 30, 20, 1.85),
        (40, O.FLT2): (30, 0.37, 40, 1, 1.94), (40, O.SMO1): (0, 0.0, 105, 105, 2.4),
    }
    for (s, m), (nx, bx, nt, na, bt) in want.items():
        p = O.default_params(s, m)
        assert (p.patch_sz, p.search_sz_x, p.search_sz_t) == (8, 10, 5)
        assert (p.npatches_x, p.npatches_t, p.npatches_tagg) == (nx, nt, na)
        assert abs(p.beta_x - bx) < 1e-6 and abs(p.beta_t - bt) < 1e-6
        assert p.dista_lambda == 1.0
    # user-set fields are kept, small sigma clamps (max(5, .), max(1, .))
    p = O.default_params(2.0, O.SMO1, patch_sz=12, npatches_t=7)
    assert (p.patch_sz, p.npatches_t, p.npatches_tagg) == (12, 7, 7)
    assert O.default_params(2.0, O.FLT2).npatches_t == 5
    assert O.default_params(60.0, O.SMO1).beta_t == 1.0


def test_window_known_values(O):
    """reference: src/nlkalman.c:401-407; psz=8 corner / centre."""
    W = O.window(8)
    assert abs(W[0, 0] - 0.0019305) < 1e-7
    assert abs(W[3, 3] - 0.8802485) < 1e-6
    assert np.allclose(W, W.T) and np.allclose(W, W[::-1, ::-1])
    assert np.array_equal(W, ref_numpy.window(8))
    assert np.allclose(O.window(12), ref_numpy.window(12), rtol=0, atol=1e-7)


def test_dct_is_orthonormal_dct2(O):
    """FFTW REDFT10 x the reference's scaling == orthonormal DCT-II."""
    rng = np.random.default_rng(0)
    for n in (4, 6, 8, 10, 12, 16):
        x = rng.normal(0, 50, (5, n, n)).astype(np.float32)
        y = O.dct2(x)
        ref = dctn(x.astype(np.float64), type=2, norm="ortho", axes=(1, 2))
        assert np.abs(y - ref).max() < 1e-4 * n
        back = O.dct2(y, inverse=True)
        assert np.abs(back - x).max() < 1e-4 * n
        assert np.abs(back - idctn(y.astype(np.float64), type=2, norm="ortho", axes=(1, 2))).max() < 1e-4 * n
        C = O.dct_basis(n).astype(np.float64)
        assert np.abs(C @ C.T - np.eye(n)).max() < 1e-6


def test_dct_ramp_known_answer(O):
    """8x8 ramp x + 8y: DC = mean * 8, only first row/column non-zero."""
    x = (np.arange(8)[None, :] + 8.0 * np.arange(8)[:, None]).astype(np.float32)
    y = O.dct2(x[None])[0]
    assert abs(y[0, 0] - 31.5 * 8) < 1e-3
    assert np.abs(y[1:, 1:]).max() < 1e-3
    assert abs(y[0, 1] - (-18.2216)) < 1e-3 and abs(y[1, 0] - 8 * (-18.2216)) < 1e-2


def test_colour_transform(O):
    rng = np.random.default_rng(1)
    im = rng.uniform(0, 255, (7, 9, 3)).astype(np.float32)
    opp = O.rgb2opp(im)
    assert np.allclose(opp[..., 0], im.sum(2) / np.sqrt(3), atol=1e-3)
    assert np.allclose(opp[..., 1], (im[..., 0] - im[..., 2]) / np.sqrt(2), atol=1e-3)
    assert np.allclose(opp[..., 2], (im[..., 0] - 2 * im[..., 1] + im[..., 2]) / np.sqrt(6), atol=1e-3)
    assert np.abs(O.opp2rgb(opp) - im).max() < 1e-3
    gray = rng.uniform(0, 255, (5, 5, 1)).astype(np.float32)
    assert np.array_equal(O.rgb2opp(gray), gray)  # no-op unless ch == 3


def test_warp_nan_ring_and_ramp(O):
    """reference: src/nlkalman.c:29-88 — zero flow gives a NaN ring of 1 px
    (left/top) and 2 px (right/bottom); a linear ramp is reproduced exactly."""
    h, w = 6, 8
    im = (np.arange(w)[None, :] * 3.0 + np.arange(h)[:, None] * 5.0).astype(np.float32)[:, :, None]
    out = O.warp_bicubic(im, np.zeros((h, w, 2), np.float32))
    nan = np.isnan(out[..., 0])
    want = np.ones((h, w), bool)
    want[1:h - 2, 1:w - 2] = False
    assert np.array_equal(nan, want)
    assert np.array_equal(out[~nan], im[~nan])
    flow = np.zeros((h, w, 2), np.float32)
    flow[..., 0], flow[..., 1] = 0.5, -0.25
    out = O.warp_bicubic(im, flow)
    ok = ~np.isnan(out[..., 0])
    assert ok.sum() > 0
    assert np.abs(out[ok] - (im + 1.5 - 1.25)[ok]).max() < 1e-4
    occ = np.zeros((h, w), np.float32)
    occ[3, 4] = 255
    assert np.isnan(O.warp_bicubic(im, np.zeros((h, w, 2), np.float32), occ)[3, 4, 0])


def test_awgn_generator(O, synth):
    """LCG + Box-Muller of lib/imscript-lite/src/random.c: numpy jump-ahead
    version == sequential C version; first LCG output for seed 0 is known."""
    assert int(synth.lcg_stream(1, 0)[0]) == (1442695040888963407 >> 32)
    c = synth.clean_frame(40, 30, 3)
    a, b = synth.awgn(c, 20, 5), O.awgn(c, 20, 5)
    assert np.abs(a - b).max() < 1e-4
    assert abs((a - c).std() - 20) < 1.0


# ------------------------------------------------------------ golden fixtures

@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_golden(O, name):
    out = cases.run_chain(O, name)
    with np.load(os.path.join(GOLD, name + ".npz")) as g:
        for k in g.files:
            cases.assert_close(out[k], g[k], f"{name}/{k}", maxabs=1e-5, rmse=1e-6)
    assert not np.isnan(out["f2_1"]).any()
    I = cases.inputs(name)
    assert cases.synth.psnr(out["rgb_f2_1"], I["clean1"]) > cases.synth.psnr(I["n1"], I["clean1"]) + 6


@pytest.mark.parametrize("name", ["gray64_s20", "rgb72x48_s40"])
def test_oracle_vs_survey_shim_build(O, name):
    """Supplementary (not a pin): outputs of the reference's own sources built at
    survey time against a DCT stand-in (tools/make_survey_fixtures.py). Each
    oracle stage is fed the survey build's previous-stage files, as the CLI does."""
    I = cases.inputs(name)
    s = I["sigma"]
    with np.load(os.path.join(GOLD, "survey_shim_" + name + ".npz")) as g:
        R = {k: g[k] for k in g.files}
    p1, p2, ps = (O.default_params(s, m) for m in (O.FLT1, O.FLT2, O.SMO1))
    o0, o1 = O.rgb2opp(I["n0"]), O.rgb2opp(I["n1"])
    f1_0 = O.filter_frame(o0, None, None, s, p1)
    cases.assert_close(O.opp2rgb(f1_0), R["f1_0"], "f1_0", maxabs=1e-3)
    cases.assert_close(O.opp2rgb(O.filter_frame(o0, None, f1_0, s, p2)), R["f2_0"], "f2_0", maxabs=1e-3)
    w1 = O.warp_bicubic(O.rgb2opp(R["f1_0"]), I["flow"], I["occ"])
    w2 = O.warp_bicubic(O.rgb2opp(R["f2_0"]), I["flow"], I["occ"])
    f1_1 = O.filter_frame(o1, w1, None, s, p1)
    cases.assert_close(O.opp2rgb(f1_1), R["f1_1"], "f1_1", maxabs=1e-3)
    cases.assert_close(O.opp2rgb(O.filter_frame(o1, w2, f1_1, s, p2)), R["f2_1"], "f2_1", maxabs=1e-3)
    ws = O.warp_bicubic(O.rgb2opp(R["f2_1"]), -I["flow"], I["occ"])
    s1 = O.smooth_frame(O.rgb2opp(R["f2_0"]), ws, None, s, ps)
    cases.assert_close(O.opp2rgb(s1), R["s1_0"], "s1_0", maxabs=1e-3)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_against_the_reference_filter(O, name):
    """THE pin the oracle lacks: src/nlkalman.c itself, compiled unmodified against a real FFTW3
    by `make -C oracle ref` where one exists (oracle/Makefile: _ref/libnlkalman_ref.so, serial build). This image
    has no FFTW: the test skips, and every filter parity statement stays "unpinned". Where the library exists,
    every stage of the golden chain must agree with the reference to the FP noise floor of two DCT
    implementations (max-abs 2e-4, RMSE 2.5e-5)."""
    import ctypes as C
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libnlkalman_ref.so")
    if not os.path.exists(so):
        pytest.skip("no reference filter build (FFTW3 absent from this image): parity unpinned")
    R = C.CDLL(so)
    fp = C.POINTER(C.c_float)

    class Ref:   # the same function names as the oracle module, backed by the reference's own code
        FLT1, FLT2, SMO1 = 0, 1, 2
        default_params = staticmethod(O.default_params)
        rgb2opp, opp2rgb, warp_bicubic = staticmethod(O.rgb2opp), staticmethod(O.opp2rgb), staticmethod(O.warp_bicubic)

        @staticmethod
        def _frame(fn, cur, prev, basic, sigma, p):
            cur = np.ascontiguousarray(cur, np.float32)
            out = np.empty_like(cur)
            h, w, ch = cur.shape
            ptr = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data_as(fp)
            keep = [np.ascontiguousarray(a, np.float32) if a is not None else None for a in (prev, basic)]
            fn.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, type(p), C.c_int]
            fn.restype = None
            fn(out.ctypes.data_as(fp), cur.ctypes.data_as(fp), ptr(keep[0]), ptr(keep[1]), w, h, ch, C.c_float(sigma), p, 0)
            return out

        @classmethod
        def filter_frame(cls, cur, prev, basic, sigma, p):
            return cls._frame(R.nlkalman_filter_frame, cur, prev, basic, sigma, p)

        @classmethod
        def smooth_frame(cls, cur, prev, basic, sigma, p):
            return cls._frame(R.nlkalman_smooth_frame, cur, prev, basic, sigma, p)
    ref = cases.run_chain(Ref, name)
    got = cases.run_chain_stagewise(O, ref, name)
    for k in ("f1_0", "f2_0", "f1_1", "f2_1", "s1_0"):
        cases.assert_close(got[k], ref[k], f"oracle vs reference filter, {name}/{k}", maxabs=1e-3, rmse=1e-4)


# ------------------------------------------- independent numpy restatement

def _small(synth, w, h, ch, sigma, seed):
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, seed)
    return n0, n1


@pytest.mark.parametrize("w,h,ch", [(28, 24, 1), (24, 20, 3)])
def test_oracle_vs_numpy_restatement(O, synth, w, h, ch):
    sigma = 20.0
    n0, n1 = _small(synth, w, h, ch, sigma, 11)
    o0, o1 = O.rgb2opp(n0), O.rgb2opp(n1)
    over = dict(search_sz_x=6, npatches_x=20, npatches_t=12, npatches_tagg=6)
    p1 = O.default_params(sigma, O.FLT1, **over)
    p2 = O.default_params(sigma, O.FLT2, **over)
    ps = O.default_params(sigma, O.SMO1, npatches_t=12)
    a = O.filter_frame(o0, None, None, sigma, p1)
    cases.assert_close(a, ref_numpy.frame(o0, None, None, sigma, p1.as_dict()), "flt1-x", maxabs=2e-3)
    prev = a.copy()
    prev[5:9, 7:12, :] = np.nan          # hole
    prev[:, 0, :] = np.nan               # border column
    b = O.filter_frame(o1, prev, None, sigma, p1)
    cases.assert_close(b, ref_numpy.frame(o1, prev, None, sigma, p1.as_dict()), "flt1-t", maxabs=2e-3)
    c = O.filter_frame(o1, prev, b, sigma, p2)
    cases.assert_close(c, ref_numpy.frame(o1, prev, b, sigma, p2.as_dict()), "flt2-t", maxabs=2e-3)
    d = O.smooth_frame(a, prev, None, sigma, ps)
    cases.assert_close(d, ref_numpy.frame(a, prev, None, sigma, ps.as_dict(), smoother=True), "smo1",
                       maxabs=2e-3)


def test_oracle_openmp_close_to_serial(O, synth):
    """Thread-order perturbation of the processed mask moves PSNR by ~0.001 dB
    (SURVEY.md §8 note N1); the parallel mode is what bench.py times."""
    n0, n1, c1 = synth.noisy_pair(128, 96, 1, 20.0, 4)
    p = O.default_params(20.0, O.FLT1)
    ser = O.filter_frame(n1, None, None, 20.0, p, nthreads=1)
    par = O.filter_frame(n1, None, None, 20.0, p, nthreads=4)
    assert abs(synth.psnr(ser, c1) - synth.psnr(par, c1)) < 0.05
    ser2 = O.filter_frame(n1, None, None, 20.0, p, nthreads=1)
    assert np.array_equal(ser, ser2)  # serial order is deterministic


def test_oracle_edge_cases(O):
    """Single-target image, k larger than the window, all-NaN previous frame."""
    rng = np.random.default_rng(5)
    im = rng.uniform(0, 255, (8, 8, 1)).astype(np.float32)
    p = O.default_params(20.0, O.FLT1)
    out, tr = O.filter_frame(im, None, None, 20.0, p, trace=True)
    assert tr["grid"] == (1, 1) and tr["nsel"][0] == 1 and tr["nagg"][0] == 1
    assert np.isfinite(out).all()
    im = rng.uniform(0, 255, (20, 16, 3)).astype(np.float32)
    prev = np.full_like(im, np.nan)
    out, tr = O.filter_frame(im, prev, None, 20.0, p, trace=True)
    assert (tr["np0"] == 0).all() and tr["active"].all()   # spatial branch, mask never marked
    out2 = O.filter_frame(im, None, None, 20.0, p)
    assert not np.array_equal(out, out2)                    # (no skip vs skip)
    ps = O.default_params(20.0, O.SMO1)
    sm = O.smooth_frame(im, prev, None, 20.0, ps)
    assert np.abs(sm - im).max() < 1e-3                     # pass-through
    flat = np.full((24, 24, 1), 100.0, np.float32)          # every distance ties at 0
    out, tr = O.filter_frame(flat, None, None, 20.0, p, trace=True)
    assert np.abs(out - 100).max() < 1e-3
    t0 = tr["topk"][0][:tr["nsel"][0]]
    assert (t0[:5] == np.array([0, 1, 2, 3, 4])).all()      # ties keep raster (window) order


# ------------------------------------------------------------ the row formulation of the mask replay

def _rows_replay_words(marks, ngx, ngy):
    """numpy restatement of k_mask_commit_rows1 (csrc/k_commit.h), word for word: 32 columns per uint32
    word, run-start / add-carry parity trick per word, carry-in across words with the all-ones fix-up."""
    nw = (ngx + 31) // 32
    fwd = (marks.reshape(ngy, ngx) >> np.uint64(5)).astype(np.uint32)   # reach 1: bits after the centre bit

    def planes(j):
        out = np.zeros((4, nw), np.uint32)
        for p in range(4):
            bits = (fwd[j] >> np.uint32(p)) & np.uint32(1)
            for i in np.nonzero(bits)[0]:
                out[p, i >> 5] |= np.uint32(1) << np.uint32(i & 31)
        return out

    colmask = np.zeros(nw, np.uint32)
    for wd in range(nw):
        nb = ngx - 32 * wd
        colmask[wd] = 0xFFFFFFFF if nb >= 32 else (1 << nb) - 1
    E, Od, ONES = np.uint32(0x55555555), np.uint32(0xAAAAAAAA), np.uint32(0xFFFFFFFF)
    a = np.zeros(nw, np.uint32)
    act = np.zeros((ngy, ngx), np.uint8)
    with np.errstate(over="ignore"):
        for j in range(ngy):
            ms, ml, md, mr = planes(j)
            g = ms & ~a
            sw = g & ~(g << np.uint32(1))
            er = g & ~(g + (sw & E))
            c0 = g & ((er & E) | (~er & Od))
            cout = c0 >> np.uint32(31)
            for wd in np.nonzero(g == ONES)[0]:          # ascending: a word of ones hands its carry-in on
                cout[wd] = cout[wd - 1] if wd else 0
            cin = np.concatenate(([0], cout[:-1])).astype(np.uint32)
            low = g & ~(g + np.uint32(1))
            c = c0 ^ (low & (np.uint32(0) - cin))
            cl = (c << np.uint32(1)) | cin
            x = ~(a | cl) & colmask
            for i in range(ngx):
                act[j, i] = (x[i >> 5] >> np.uint32(i & 31)) & 1
            xl, xd, xr = x & ml, x & md, x & mr
            nxt = np.concatenate((xl[1:], [0])).astype(np.uint32)
            prv = np.concatenate(([0], xr[:-1])).astype(np.uint32)
            a = ((xl >> np.uint32(1)) | (nxt << np.uint32(31))) | xd | ((xr << np.uint32(1)) | (prv >> np.uint32(31)))
    return act.ravel()


def test_row_formulation_of_the_mask_replay(O):
    """The GPU replays the reach-1 processed mask one grid ROW per step as a word-parallel carry chain
    (k_mask_commit_rows1). Its arithmetic, restated on uint32 words, must give the serial loop's decisions
    (reference: src/nlkalman.c:597-600, 930-931) — also for rows whose marks fill whole words."""
    rng = np.random.default_rng(5)
    for ngx, ngy in ((1, 1), (31, 4), (32, 6), (33, 9), (64, 3), (97, 12), (130, 5)):
        for density, right in ((0.5, 0.5), (0.9, 0.97), (0.15, 1.0), (1.0, 1.0)):
            bits = rng.random((ngx * ngy, 9)) < density
            bits[:, 5] = rng.random(ngx * ngy) < right          # (di = +1, dj = 0)
            if right == 1.0 and density < 1.0:
                bits[:, 6:] &= (rng.random((ngx * ngy, 1)) < 0.3)
            marks = (bits.astype(np.uint64) << np.arange(9, dtype=np.uint64)).sum(axis=1).astype(np.uint64)
            want = O.mask_commit(marks, ngx, ngy, 1)
            got = _rows_replay_words(marks, ngx, ngy)
            assert np.array_equal(got, want), (ngx, ngy, density, right)


def _rows_replay_iterated(marks, ngx, ngy, R):
    """k_mask_commit_rows<R> on numpy uint32 words (k_commit.h): marks from above as plane ANDs and whole-row
    shifts, the in-row chain by iteration to the fixed point."""
    side = 2 * R + 1
    centre = R * side + R
    nw = (ngx + 31) // 32
    U = np.uint32
    fwd = (marks >> np.uint64(centre + 1)).astype(np.uint64).reshape(ngy, ngx)

    def plane(j, p):
        b = ((fwd[j] >> np.uint64(p)) & np.uint64(1)).astype(np.uint64)
        out = np.zeros(nw, np.uint64)
        np.bitwise_or.at(out, np.arange(ngx) >> 5, b << (np.arange(ngx, dtype=np.uint64) & np.uint64(31)))
        return out.astype(U)

    def up(v, d):
        prev = np.concatenate(([0], v[:-1])).astype(U)
        return ((v << U(d)) | (prev >> U(32 - d))).astype(U)

    def down(v, d):
        nxt = np.concatenate((v[1:], [0])).astype(U)
        return ((v >> U(d)) | (nxt << U(32 - d))).astype(U)

    colmask = np.array([0xFFFFFFFF if ngx - 32 * k >= 32 else (1 << (ngx - 32 * k)) - 1 for k in range(nw)], U)
    A = [np.zeros(nw, U) for _ in range(R)]
    act = np.zeros((ngy, ngx), np.uint8)
    most = 0
    for j in range(ngy):
        D = [plane(j, p) for p in range(R + R * side)]
        n = ~A[0] & colmask
        x = n.copy()
        it = 0
        while True:
            blocked = np.zeros(nw, U)
            for d in range(1, R + 1):
                blocked |= up(x & D[d - 1], d)
            y = n & ~blocked
            it += 1
            if np.array_equal(x, y):
                break
            x = y
        most = max(most, it)
        act[j] = (x[np.arange(ngx) >> 5] >> (np.arange(ngx, dtype=U) & U(31))) & 1
        new = []
        for dj in range(1, R + 1):
            cb = x & D[R + (dj - 1) * side + R]
            for di in range(1, R + 1):
                cb = cb | up(x & D[R + (dj - 1) * side + R + di], di) | down(x & D[R + (dj - 1) * side + R - di], di)
            new.append((A[dj] if dj < R else np.zeros(nw, U)) | cb)
        A = new
    return act.ravel(), most


@pytest.mark.parametrize("reach", [1, 2, 3])
def test_iterated_row_formulation_of_the_mask_replay(O, reach):
    """Reach 2 and 3 (spatial first frames, 12 x 12 patches) replay the mask by grid rows with the in-row
    chain x_i = n_i & !(x_{i-1} & m1_{i-1}) & .. solved by iteration (k_mask_commit_rows<R>): a triangular
    system, so the fixed point is the serial loop's answer (reference: src/nlkalman.c:597-600, 930-931), reached
    within ngx + 1 iterations."""
    rng = np.random.default_rng(50 + reach)
    side = 2 * reach + 1
    c = reach * side + reach
    for ngx, ngy in ((1, 1), (31, 4), (32, 6), (33, 9), (64, 5), (97, 12), (130, 7)):
        for density, right in ((0.5, 0.5), (0.9, 0.97), (0.15, 1.0), (1.0, 1.0), (0.05, 0.3)):
            bits = rng.random((ngx * ngy, side * side)) < density
            bits[:, c + 1] = rng.random(ngx * ngy) < right
            if right == 1.0 and density < 1.0:
                bits[:, c + 2:] &= (rng.random((ngx * ngy, 1)) < 0.3)
            marks = (bits.astype(np.uint64) << np.arange(side * side, dtype=np.uint64)).sum(axis=1).astype(np.uint64)
            want = O.mask_commit(marks, ngx, ngy, reach)
            got, most = _rows_replay_iterated(marks, ngx, ngy, reach)
            assert np.array_equal(got, want), (ngx, ngy, density, right)
            assert most <= ngx + 1


def test_image_smaller_than_a_patch_comes_back_unchanged(O):
    """reference: src/nlkalman.c:586-595 (`px < w - psz + 1`: no target), :939-942 (unaggregated pixels keep the input)"""
    rng = np.random.default_rng(2)
    for (w, h, ch), psz in [((5, 20, 3), 8), ((20, 7, 1), 8), ((7, 7, 1), 8), ((11, 30, 3), 12)]:
        im = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        prev = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        p = O.default_params(20.0, O.FLT1, patch_sz=psz)
        assert np.array_equal(O.filter_frame(im, prev, None, 20.0, p), im)
        assert np.array_equal(O.filter_frame(im, None, None, 20.0, p), im)
        ps = O.default_params(20.0, O.SMO1, patch_sz=psz)
        assert np.array_equal(O.smooth_frame(im, prev, None, 20.0, ps), im)
        assert O.grid_shape(w, h, psz)[0] * O.grid_shape(w, h, psz)[1] == 0
