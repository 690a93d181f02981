"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI libraries,
against the CPU oracle on the same seeded inputs and against the committed
golden fixtures.

Tolerance (0..255 scale, opponent colour space): max-abs 2e-3, RMSE 2e-4 per
stage — ten times the path's own FP noise floor of 2e-4 / 2.5e-5,
and |dPSNR| <= 0.02 dB (BASELINE.json). Integer records (k-NN lists, group
coordinates, np0, processed-mask decisions) must match EXACTLY."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _to_o(O, p):
    return O.Params(*[getattr(p, k) for k, _ in p._fields_])


def _dev_frame(ctx, smoother, cur, prev, basic, sigma, p):
    h, w, ch = cur.shape
    d = [ctx.upload(a) if a is not None else None for a in (cur, prev, basic)]
    d_out = ctx.alloc(cur.nbytes)
    fn = ctx.smooth_frame if smoother else ctx.filter_frame
    fn(d_out, d[0], d[1], d[2], w, h, ch, sigma, p)
    out = ctx.download(d_out, cur.shape)
    rec = ctx.read_records()
    for x in d + [d_out]:
        if x:
            ctx.free(x)
    return out, rec


def _check_records(rec, tr, what):
    act_o, act_g = tr["active"].astype(bool), rec["active"].astype(bool)
    assert np.array_equal(act_o, act_g), f"{what}: processed-mask decisions differ"
    a = act_o
    for f in ("nsel", "np0", "nagg"):
        assert np.array_equal(tr[f][a], rec[f][a]), f"{what}: {f} differs"
    k = tr["topk"].shape[1]
    col = np.arange(k)[None, :]
    live = a[:, None] & (col < tr["nsel"][:, None])
    assert np.array_equal(tr["topk"][live], rec["topk"][:, :k].astype(np.int64)[live]), f"{what}: k-NN lists differ"
    g = tr["gcoords"].shape[1]
    col = np.arange(g)[None, :]
    live = a[:, None] & (col < tr["nagg"][:, None])
    assert np.array_equal(tr["gcoords"][live], rec["gcoords"][:, :g].astype(np.int64)[live]), f"{what}: group members differ"


@pytest.mark.parametrize("name", list(cases.CASES))
def test_pipeline_stagewise_vs_oracle_and_golden(built, O, name):
    """Every stage of the 2-frame flt1 -> flt2 -> smo1 pipeline, through the
    drop-in C API (libnlkalman.so), fed the oracle's previous-stage outputs."""
    ref = cases.run_chain(O, name)
    got = cases.run_chain_stagewise(built, ref, name)
    for k in ("f1_0", "f2_0", "w1", "w2", "f1_1", "f2_1", "ws", "s1_0", "rgb_f2_1"):
        cases.assert_close(got[k], ref[k], f"{name}/{k}")
    with np.load(os.path.join(GOLD, name + ".npz")) as g:
        for k in g.files:
            cases.assert_close(got[k], g[k], f"golden {name}/{k}")


@pytest.mark.parametrize("name", ["rgb96x64_s20", "gray70x53_ragged"])
def test_pipeline_end_to_end_psnr(built, O, name):
    """Free-running GPU pipeline (its own outputs feed the next stage)."""
    ref, got = cases.run_chain(O, name), cases.run_chain(built, name)
    clean = cases.inputs(name)["clean1"]
    d = cases.synth.psnr(got["rgb_f2_1"], clean) - cases.synth.psnr(ref["rgb_f2_1"], clean)
    assert abs(d) <= 0.02, d
    cases.assert_close(got["s1_0"], ref["s1_0"], "s1_0 end-to-end", maxabs=5e-3, rmse=5e-4)


@pytest.mark.parametrize("name,mode", [("rgb96x64_s20", "x"), ("rgb96x64_s20", "t"),
                                       ("gray64_s20", "t2"), ("rgb84x60_p12_s40", "t"),
                                       ("rgb72x48_s40", "s"), ("rgb40x40_p4", "x")])
def test_integer_records_exact(ctx, built, O, name, mode):
    """k-NN lists, group membership, np0 and mask decisions, bit for bit."""
    I = cases.inputs(name)
    s, over = I["sigma"], I["over"]
    ref = cases.run_chain(O, name)
    o0, o1 = O.rgb2opp(I["n0"]), O.rgb2opp(I["n1"])
    if mode == "x":
        p, args, smo = built.default_params(s, built.FLT1, **over), (o0, None, None), False
    elif mode == "t":
        p, args, smo = built.default_params(s, built.FLT1, **over), (o1, ref["w1"], None), False
    elif mode == "t2":
        p, args, smo = built.default_params(s, built.FLT2, **over), (o1, ref["w2"], ref["f1_1"]), False
    else:
        p, args, smo = built.default_params(s, built.SMO1), (ref["f2_0"], ref["ws"], None), True
    fn = O.smooth_frame if smo else O.filter_frame
    r, tr = fn(*args, s, _to_o(O, p), trace=True)
    g, rec = _dev_frame(ctx, smo, *args, s, p)
    _check_records(rec, tr, f"{name}/{mode}")
    cases.assert_close(g, r, f"{name}/{mode}")


def test_edge_cases(ctx, built, O):
    rng = np.random.default_rng(5)
    p = built.default_params(20.0, built.FLT1)
    po = _to_o(O, p)
    # single target, window = 1 candidate, k clamps to 1
    im = rng.uniform(0, 255, (8, 8, 1)).astype(np.float32)
    g, rec = _dev_frame(ctx, False, im, None, None, 20.0, p)
    cases.assert_close(g, O.filter_frame(im, None, None, 20.0, po), "8x8")
    # all-NaN previous frame: spatial branch everywhere, mask never marked
    im = rng.uniform(0, 255, (20, 16, 3)).astype(np.float32)
    prev = np.full_like(im, np.nan)
    r, tr = O.filter_frame(im, prev, None, 20.0, po, trace=True)
    g, rec = _dev_frame(ctx, False, im, prev, None, 20.0, p)
    _check_records(rec, tr, "all-NaN prev")
    cases.assert_close(g, r, "all-NaN prev")
    ps = built.default_params(20.0, built.SMO1)
    g, _ = _dev_frame(ctx, True, im, prev, None, 20.0, ps)
    cases.assert_close(g, O.smooth_frame(im, prev, None, 20.0, _to_o(O, ps)), "smoother pass-through")
    # flat image: every distance ties at zero -> raster tie-break must match
    flat = np.full((24, 24, 1), 100.0, np.float32)
    r, tr = O.filter_frame(flat, None, None, 20.0, po, trace=True)
    g, rec = _dev_frame(ctx, False, flat, None, None, 20.0, p)
    _check_records(rec, tr, "flat")
    cases.assert_close(g, r, "flat")
    # single NaN pixel in the previous frame + odd sizes + other patch sizes
    for psz, (w, h, ch) in [(6, (37, 29, 1)), (10, (45, 33, 3)), (16, (50, 40, 1)), (8, (33, 47, 3))]:
        im = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        prev = im + rng.normal(0, 5, im.shape).astype(np.float32)
        prev[h // 2, w // 2, 0] = np.nan
        pp = built.default_params(20.0, built.FLT1, patch_sz=psz, search_sz_x=6 if psz == 6 else 10)
        r, tr = O.filter_frame(im, prev, None, 20.0, _to_o(O, pp), trace=True)
        g, rec = _dev_frame(ctx, False, im, prev, None, 20.0, pp)
        _check_records(rec, tr, f"psz{psz}")
        cases.assert_close(g, r, f"psz{psz}")


def test_image_smaller_than_a_patch_is_returned_unchanged(ctx, built, O):
    """No target fits an image narrower or lower than a patch: the reference's loops `px < w - psz + 1`
    (src/nlkalman.c:586-595, :1477-1486) do not run, nothing is aggregated and every pixel keeps its input value
    (:939-942, :1853-1856). Device call, host-pointer call and oracle, filter and smoother."""
    rng = np.random.default_rng(11)
    for (w, h, ch), psz in [((5, 20, 3), 8), ((20, 7, 1), 8), ((7, 7, 3), 8), ((30, 11, 3), 12), ((3, 2, 1), 4)]:
        im = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        prev = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        for mode, smo in ((built.FLT1, False), (built.FLT2, False), (built.SMO1, True)):
            p = built.default_params(20.0, mode, patch_sz=psz)
            bas = prev if mode == built.FLT2 else None
            r = (O.smooth_frame if smo else O.filter_frame)(im, prev, bas, 20.0, _to_o(O, p))
            assert np.array_equal(r, im)
            g, _ = _dev_frame(ctx, smo, im, prev, bas, 20.0, p)
            assert np.array_equal(g, im), (w, h, ch, psz, mode)
            gh = (built.smooth_frame if smo else built.filter_frame)(im, prev, bas, 20.0, p)
            assert np.array_equal(gh, im), (w, h, ch, psz, mode)


@pytest.mark.parametrize("psz,ch,size", [(17, 1, (75, 61)), (20, 3, (90, 70)), (25, 3, (83, 77)), (32, 1, (100, 90)),
                                         (32, 3, (96, 80)), (8, 2, (60, 50)), (12, 4, (70, 64)), (7, 2, (41, 37))])
def test_large_patches_and_other_channel_counts(ctx, built, O, psz, ch, size):
    """Patch sizes 17..32 (`k_bm_generic` + `k_group_any`) and channel counts other than 1 and 3 (`k_bm_generic` +
    `k_groupp`) - the reference takes any: src/nlkalman.c:524-525, 555-560: FLT1 spatial, FLT1 temporal with a NaN hole in the previous
    frame, FLT2 and the smoother against the serial oracle - records exact, pixels within tolerance."""
    rng = np.random.default_rng(100 * psz + ch)
    w, h = size
    sigma = 20.0
    clean = np.add.outer(np.linspace(30, 200, h), np.linspace(0, 40, w))[..., None] * np.ones(ch)
    n0 = (clean + rng.normal(0, sigma, clean.shape)).astype(np.float32)
    n1 = (clean + rng.normal(0, sigma, clean.shape)).astype(np.float32)
    p1 = built.default_params(sigma, built.FLT1, patch_sz=psz)
    p2 = built.default_params(sigma, built.FLT2, patch_sz=psz)
    ps = built.default_params(sigma, built.SMO1, patch_sz=psz)
    r0, t0 = O.filter_frame(n0, None, None, sigma, _to_o(O, p1), trace=True)
    g0, rec0 = _dev_frame(ctx, False, n0, None, None, sigma, p1)
    _check_records(rec0, t0, "spatial")
    cases.assert_close(g0, r0, f"psz {psz} ch {ch}: flt1 spatial")
    prev = r0.copy()
    prev[h // 3:h // 3 + 5, w // 2:w // 2 + 9] = np.nan
    r1, t1 = O.filter_frame(n1, prev, None, sigma, _to_o(O, p1), trace=True)
    g1, rec1 = _dev_frame(ctx, False, n1, prev, None, sigma, p1)
    _check_records(rec1, t1, "temporal")
    cases.assert_close(g1, r1, f"psz {psz} ch {ch}: flt1 temporal")
    r2, t2 = O.filter_frame(n1, prev, r1, sigma, _to_o(O, p2), trace=True)
    g2, rec2 = _dev_frame(ctx, False, n1, prev, r1, sigma, p2)
    _check_records(rec2, t2, "flt2")
    cases.assert_close(g2, r2, f"psz {psz} ch {ch}: flt2")
    rs, ts = O.smooth_frame(r0, prev, None, sigma, _to_o(O, ps), trace=True)
    gs, recs = _dev_frame(ctx, True, r0, prev, None, sigma, ps)
    _check_records(recs, ts, "smo")
    cases.assert_close(gs, rs, f"psz {psz} ch {ch}: smo1")
    # ... and through the drop-in API (host pointers)
    cases.assert_close(built.filter_frame(n1, prev, None, sigma, p1), r1, f"psz {psz} ch {ch}: host call")


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_warp_bicubic_any_channel_count(built, O, ch):
    """`warp_bicubic` (src/nlkalman.c:29-88) through the drop-in API for 1..4 channels - `k_warp_bicubic<1>`, `<3>`
    and the tap-by-tap kernel for the rest: flows that leave the image (NaN ring), an occlusion mask, bit-exact
    against the oracle (same mixed float / double evaluation)."""
    rng = np.random.default_rng(40 + ch)
    h, w = 57, 83
    im = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
    im[10:12, 20:23] = np.nan
    flow = rng.normal(0, 3, (h, w, 2)).astype(np.float32)
    flow[:5] += 9.0
    occ = (rng.random((h, w)) < 0.05).astype(np.float32)
    for m in (None, occ):
        r, g = O.warp_bicubic(im, flow, m), built.warp_bicubic(im, flow, m)
        assert np.array_equal(np.isnan(r), np.isnan(g))
        assert np.array_equal(np.nan_to_num(r), np.nan_to_num(g)), ch


def test_unsupported_parameters_fail_loudly(ctx, built):
    """What is left outside the kernels: patches above 32x32 (or ch * psz^2 > 4096), and - across GPUs only - a
    marking group that reaches more than 3 grid cells (64-bit mark words)."""
    im = np.zeros((40, 40, 1), np.float32)
    d, o = ctx.upload(im), ctx.alloc(im.nbytes)
    with pytest.raises(built.NlkError, match="not supported"):
        ctx.filter_frame(o, d, None, None, 40, 40, 1, 20.0, built.default_params(20.0, 0, patch_sz=34))
    im5 = np.zeros((40, 40, 5), np.float32)
    d5, o5 = ctx.upload(im5), ctx.alloc(im5.nbytes)
    with pytest.raises(built.NlkError, match="not supported"):
        ctx.filter_frame(o5, d5, None, None, 40, 40, 5, 20.0, built.default_params(20.0, 0, patch_sz=30))
    ctx.free(d5); ctx.free(o5)
    marks = ctx.upload(np.zeros(19 * 19, np.uint64))
    with pytest.raises(built.NlkError, match="reach"):
        ctx.strip_match(marks, d, None, None, 40, 40, 1, 20.0, built.default_params(20.0, 0, patch_sz=4, search_sz_x=10), 0, 19)
    for x in (d, o, marks):
        ctx.free(x)


def test_strip_accumulate_equals_whole_frame(ctx, built):
    """Row-strip form: accumulating the grid rows in two calls into one
    accumulator, then normalising, reproduces the whole-frame call when the
    processed-mask cannot couple the strips (FLT2: npatches_tagg = 1)."""
    I = cases.inputs("rgb96x64_s20")
    s = I["sigma"]
    w, h, ch = I["w"], I["h"], I["ch"]
    o1 = built.rgb2opp(I["n1"])
    p = built.default_params(s, built.FLT2)
    whole = built.filter_frame(o1, None, o1, s, p)
    d_cur = ctx.upload(o1)
    acc = ctx.upload(np.zeros((ch + 1, h, w), np.float32))
    d_out = ctx.alloc(o1.nbytes)
    step = p.patch_sz // 2
    ngy = (h - p.patch_sz) // step + 1
    half = ngy // 2
    ctx.frame_accumulate(acc, d_cur, None, d_cur, w, h, ch, s, p, 0, half)
    ctx.frame_accumulate(acc, d_cur, None, d_cur, w, h, ch, s, p, half * step, ngy - half)
    ctx.frame_normalize(d_out, acc, d_cur, w, h, ch, 0, h)
    got = ctx.download(d_out, o1.shape)
    cases.assert_close(got, whole, "two strips vs whole")
    for x in (d_cur, acc, d_out):
        ctx.free(x)


@pytest.mark.parametrize("world", [2, 3])
def test_gpu_strips_match_oracle_strips(ctx, built, O, synth, world):
    """The row-strip form exactly as the multi-GPU driver uses it (strip + halo
    images, oy / ngy target rows, halo rows of the accumulator added to the
    neighbour, per-strip mask replay) on one GPU, against the same procedure
    on the oracle; temporal FLT1 so that the mask really couples targets."""
    import importlib
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    w, h, ch, sigma = 64, 96, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 21)
    o0, o1 = O.rgb2opp(n0), O.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    po = _to_o(O, p)
    prev = O.filter_frame(o0, None, None, sigma, po)
    plan = strips.strip_plan(h, p.patch_sz, max(p.search_sz_x, p.search_sz_t), world)
    step = p.patch_sz // 2
    acc_g = np.zeros((ch + 1, h, w), np.float32)
    acc_o = np.zeros((ch + 1, h, w), np.float32)
    for s in plan:
        hl = s["Y1"] - s["Y0"]
        cur_s = np.ascontiguousarray(o1[s["Y0"]:s["Y1"]])
        prev_s = np.ascontiguousarray(prev[s["Y0"]:s["Y1"]])
        oy, ngy = s["gy0"] * step - s["Y0"], s["gy1"] - s["gy0"]
        a = np.zeros((ch + 1, hl, w), np.float32)
        O.frame_accumulate(a, cur_s, prev_s, None, sigma, po, oy, ngy)
        acc_o[:, s["Y0"]:s["Y1"]] += a
        d_cur, d_prev, d_acc = ctx.upload(cur_s), ctx.upload(prev_s), ctx.upload(np.zeros_like(a))
        ctx.frame_accumulate(d_acc, d_cur, d_prev, None, w, hl, ch, sigma, p, oy, ngy)
        acc_g[:, s["Y0"]:s["Y1"]] += ctx.download(d_acc, a.shape)
        for x in (d_cur, d_prev, d_acc):
            ctx.free(x)
    want = O.frame_normalize(acc_o, o1, 0, h)
    d_acc, d_cur, d_out = ctx.upload(acc_g), ctx.upload(o1), ctx.alloc(o1.nbytes)
    ctx.frame_normalize(d_out, d_acc, d_cur, w, h, ch, 0, h)
    got = ctx.download(d_out, o1.shape)
    for x in (d_acc, d_cur, d_out):
        ctx.free(x)
    cases.assert_close(got, want, f"{world} strips on the GPU vs on the oracle")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("world", [2, 4])
def test_gpu_exact_strips_equal_whole_frame(ctx, built, O, synth, world, fused):
    """The three-phase strip form (match per strip -> concatenated mark words ->
    whole-grid mask replay -> group per strip), i.e. what N ranks do in exact
    mode, run sequentially on one GPU: must equal the whole-frame call. `fused`: phases 2 + 3 of a strip as
    one call (nlk_dev_strip_commit_group: the replay of the rows down to the strip's last one inside the group
    kernel's launch, what csrc/strips.hip enqueues) instead of mask_commit + strip_group."""
    import importlib
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    w, h, ch, sigma = 96, 128, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 33)
    o0, o1 = O.rgb2opp(n0), O.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    prev = O.filter_frame(o0, None, None, sigma, _to_o(O, p))
    whole, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    plan = strips.strip_plan(h, p.patch_sz, max(p.search_sz_x, p.search_sz_t), world)
    step = p.patch_sz // 2
    ngx, ngy = (w - p.patch_sz) // step + 1, (h - p.patch_sz) // step + 1
    d_marks = ctx.upload(np.zeros(ngx * ngy, np.uint64))
    d_active = ctx.upload(np.zeros(ngx * ngy, np.uint8))
    bufs = []
    for s in plan:  # phase 1 on every strip
        cur_s = np.ascontiguousarray(o1[s["Y0"]:s["Y1"]])
        prev_s = np.ascontiguousarray(prev[s["Y0"]:s["Y1"]])
        d_cur, d_prev = ctx.upload(cur_s), ctx.upload(prev_s)
        oy, ngy_l = s["gy0"] * step - s["Y0"], s["gy1"] - s["gy0"]
        reach = ctx.strip_match(d_marks + 8 * s["gy0"] * ngx, d_cur, d_prev, None, w, cur_s.shape[0], ch,
                                sigma, p, oy, ngy_l)
        bufs.append((d_cur, d_prev, cur_s.shape[0], oy, ngy_l))
    ctx.mask_commit(d_marks, ngx, ngy, reach, d_active)  # phase 2 on the whole grid
    assert np.array_equal(ctx.download(d_active, (ngx * ngy,), np.uint8), rec["active"])
    acc = np.zeros((ch + 1, h, w), np.float32)
    for s, (d_cur, d_prev, hl, oy, ngy_l) in zip(plan, bufs):  # phase 3 per strip
        ctx.strip_match(None, d_cur, d_prev, None, w, hl, ch, sigma, p, oy, ngy_l)  # restore the strip state
        d_acc = ctx.upload(np.zeros((ch + 1, hl, w), np.float32))
        if fused:
            d_scratch = ctx.upload(np.full(ngx * ngy, 7, np.uint8))   # (not read: the launch replays the mask itself)
            ctx.strip_commit_group(d_acc, d_marks, ngx, ngy, reach, s["gy0"], d_scratch)
            ctx.free(d_scratch)
        else:
            ctx.strip_group(d_acc, d_active + s["gy0"] * ngx)
        acc[:, s["Y0"]:s["Y1"]] += ctx.download(d_acc, (ch + 1, hl, w))
        for x in (d_cur, d_prev, d_acc):
            ctx.free(x)
    d_acc, d_cur, d_out = ctx.upload(acc), ctx.upload(o1), ctx.alloc(o1.nbytes)
    ctx.frame_normalize(d_out, d_acc, d_cur, w, h, ch, 0, h)
    got = ctx.download(d_out, o1.shape)
    for x in (d_acc, d_cur, d_out, d_marks, d_active):
        ctx.free(x)
    cases.assert_close(got, whole, f"exact strips x{world} vs whole frame (fused {fused})", maxabs=5e-4, rmse=5e-5)


def _two_rank_worker(rank, world, port, q, backend="gloo"):
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    pkg = importlib.import_module("bwd-nlkalman_amd")
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    dev = torch.device("cuda", 0)
    if backend == "nccl":   # RCCL: the backend bench.py uses on a multi-GPU node
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w, h, ch, sigma = 320, 256, 3, 20.0
    ctx = pkg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 5)
    t_n0, t_n1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    ctx.rgb2opp(t_n0.data_ptr(), w, h, ch)
    ctx.rgb2opp(t_n1.data_ptr(), w, h, ch)
    p = pkg.default_params(sigma, pkg.FLT1)
    t_prev, t_whole = torch.empty_like(t_n0), torch.empty_like(t_n0)
    ctx.filter_frame(t_prev.data_ptr(), t_n0.data_ptr(), None, None, w, h, ch, sigma, p)
    ctx.filter_frame(t_whole.data_ptr(), t_n1.data_ptr(), t_prev.data_ptr(), None, w, h, ch, sigma, p)

    def accumulate(acc, cur, prev, oy, ngy):
        ctx.frame_accumulate(acc.data_ptr(), cur.data_ptr(), prev.data_ptr(), None, w, cur.shape[0], ch,
                             sigma, p, oy, ngy)

    def normalize(out, acc, cur, y0, y1):
        ctx.frame_normalize(out.data_ptr(), acc.data_ptr(), cur.data_ptr(), w, cur.shape[0], ch, y0, y1)

    def match(marks, cur, prev, oy, ngy):
        return ctx.strip_match(marks.data_ptr(), cur.data_ptr(), prev.data_ptr(), None, w, cur.shape[0],
                               ch, sigma, p, oy, ngy)

    def match_rows(marks, cur, prev, oy, ngy, r0, rows, lay):
        return ctx.strip_match_part(marks.data_ptr(), cur.data_ptr(), prev.data_ptr(), None, w, cur.shape[0],
                                    ch, sigma, p, oy, ngy, r0, rows, lay)

    def commit(marks_full, ngx, ngy, reach, active_full):
        ctx.mask_commit(marks_full.data_ptr(), ngx, ngy, reach, active_full.data_ptr())

    def group(acc, active):
        ctx.strip_group(acc.data_ptr(), active.data_ptr())
    res = {}
    for mode, phases in (("exact", (match, commit, group)), ("exact-overlap", (match, commit, group, match_rows)),
                         ("per-strip", None)):
        sf = strips.StripFrame(rank, world, w, h, ch, p.patch_sz, max(p.search_sz_x, p.search_sz_t), dev,
                               accumulate, normalize, phases=phases, stage_host=backend != "nccl")
        sf.load(t_n1, t_prev)
        for _ in range(2):
            # (the halo rows of the previous frame hold garbage until the exchange delivers them: a seam target
            # matched too early, or a layout that reads them in flight, would show)
            pp = sf.p
            sf.prev[:sf._l(pp["own0"])] = float("nan")
            sf.prev[sf._l(pp["own1"]):] = float("nan")
            sf.step()
        y0, y1, rows = sf.own_rows()
        full = torch.zeros((h, w, ch), device=dev if backend == "nccl" else "cpu")
        full[y0:y1] = rows if backend == "nccl" else rows.cpu()
        dist.all_reduce(full)
        res[mode] = full.cpu().numpy()
    if rank == 0:
        q.put((res, t_whole.cpu().numpy()))
    dist.destroy_process_group()


def test_two_processes_drive_gpu_strips(built):
    """bench.py's N > 1 code path end to end on the one GPU of the test box: two
    processes (gloo, exchanged tensors staged through the host) run the HIP
    kernels on their strips; exact mode must equal the whole-frame call."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res, whole = q.get(timeout=600)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    cases.assert_close(res["exact"], whole, "2 processes, exact mode, vs whole frame", maxabs=5e-4, rmse=5e-5)
    cases.assert_close(res["exact-overlap"], whole, "2 processes, interior rows matched during the halo exchange",
                       maxabs=5e-4, rmse=5e-5)
    d = np.abs(res["per-strip"] - whole)
    assert np.isfinite(res["per-strip"]).all() and (d > 1e-2).mean() < 0.2  # seam-order differences only


def test_strip_driver_over_rccl_one_rank(built):
    """The strip driver on the backend bench.py uses on a multi-GPU node ("nccl" = RCCL, device tensors, no host
    staging): process-group creation on the device, the 64-bit mark words through `all_gather_into_tensor`, the
    all-reduce of the assembled rows. One rank is all a one-GPU box can hold (RCCL refuses two ranks on one
    device), so the neighbour sends have no peer here: those run over gloo in the two-process test above."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    pr = mpc.Process(target=_two_rank_worker, args=(0, 1, port, q, "nccl"))
    pr.start()
    res, whole = q.get(timeout=600)
    pr.join(timeout=120)
    assert pr.exitcode == 0
    for mode in ("exact", "exact-overlap", "per-strip"):
        cases.assert_close(res[mode], whole, f"RCCL, one rank, {mode}", maxabs=5e-4, rmse=5e-5)


def test_full_size_1080p_against_oracle(ctx, built, O, synth):
    """BASELINE.json configs[1] at full size: 1920x1080 RGB sigma=20 FLT1
    temporal. The serial oracle needs ~25 s; mask decisions must be identical,
    pixels within tolerance (a handful of 1e-6-threshold flips allowed)."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p)
    g, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    r, tr = O.filter_frame(o1, prev, None, sigma, _to_o(O, p), trace=True)
    assert np.array_equal(tr["active"].astype(bool), rec["active"].astype(bool))
    assert 0.2 < 1 - tr["active"].mean() < 0.4          # ~30 % of targets are skipped
    _check_records(rec, tr, "1080p")
    g, _ = cases.excuse_threshold_pixels(g, r, tr, "1080p", 64)
    cases.assert_close(g, r, "1080p")
    dpsnr = synth.psnr(built.opp2rgb(g), c1) - synth.psnr(O.opp2rgb(r), c1)
    assert abs(dpsnr) <= 0.02
    # size-independent property: a DC offset on every input shifts the output by it
    off = np.float32(16.0)
    prev2 = prev.copy()
    prev2[..., 0] += off
    o2 = o1.copy()
    o2[..., 0] += off
    g2, _ = _dev_frame(ctx, False, o2, prev2, None, sigma, p)
    d = np.abs(g2[..., 0] - g[..., 0] - off)
    # (adding the offset re-rounds the pixels, so a few near-tied k-NN ranks flip: quantiles)
    assert np.quantile(d, 0.9999) < 2e-2
    assert np.quantile(np.abs(g2[..., 1:] - g[..., 1:]), 0.9999) < 2e-2


@pytest.mark.parametrize("band", [8, 11, 37])
def test_banded_mask_replay_is_exact(ctx, built, O, synth, monkeypatch, band):
    """Patch grids of more than 1024 rows (8K frames) replay the processed-mask in bands of rows
    that start with the previous band's last rows as context (k_commit.h). With the band forced
    small (NLK_COMMIT_BAND) the decisions must still be the oracle's, for the temporal reach (1),
    the spatial reach (3) and the second iteration, and equal the single-band run on a larger frame."""
    monkeypatch.setenv("NLK_COMMIT_BAND", str(band))
    for name, mode in (("rgb96x64_s20", "x"), ("rgb96x64_s20", "t"), ("gray70x53_ragged", "x")):
        I = cases.inputs(name)
        s, over = I["sigma"], I["over"]
        ref = cases.run_chain(O, name)
        o0, o1 = O.rgb2opp(I["n0"]), O.rgb2opp(I["n1"])
        p = built.default_params(s, built.FLT1, **over)
        args = (o0, None, None) if mode == "x" else (o1, ref["w1"], None)
        _, tr = O.filter_frame(*args, s, _to_o(O, p), trace=True)
        _, rec = _dev_frame(ctx, False, *args, s, p)
        _check_records(rec, tr, f"{name}/{mode}/band {band}")
    w, h, ch, sigma = 400, 360, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 11)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1)
    f0b, r0b = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    f1b, r1b = _dev_frame(ctx, False, o1, f0b, None, sigma, p1)
    monkeypatch.delenv("NLK_COMMIT_BAND")
    f0, r0 = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    f1, r1 = _dev_frame(ctx, False, o1, f0b, None, sigma, p1)
    assert np.array_equal(r0b["active"], r0["active"]) and np.array_equal(r1b["active"], r1["active"])
    assert 0.2 < 1 - r1["active"].mean() < 0.5   # (the skip really is exercised)


def test_block_matching_equals_target_by_target(ctx, built, synth, monkeypatch):
    """k_bm_topk shares the squared differences inside blocks of 4 x 2 targets (nlk_match_block);
    NLK_MATCH_NOBLOCK=1 runs every target on its own (nlk_match_target). Same per-target summation
    order, so ALL records - every target's sorted k-NN list, group, counters, mark decisions - must be
    identical, for spatial and temporal frames, with and without a basic estimate, the smoother, and
    with NaN holes."""
    w, h, ch, sigma = 1000, 560, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 21)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, p2, p3 = (built.default_params(sigma, m) for m in (built.FLT1, built.FLT2, built.SMO1))
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    holes = prev.copy()
    holes[100:140, 300:420] = np.nan
    holes[:, :2] = np.nan
    calls = [(False, o0, None, None, p1),  # (spatial: 441 candidates, seven rounds per block)
             (False, o1, prev, None, p1), (False, o1, holes, None, p1), (False, n1, prev, o1, p2),
             (True, o1, prev, None, p3), (True, o1, holes, None, p3)]
    for smo, cur, pv, basic, p in calls:
        monkeypatch.delenv("NLK_MATCH_NOBLOCK", raising=False)
        _, ra = _dev_frame(ctx, smo, cur, pv, basic, sigma, p)
        monkeypatch.setenv("NLK_MATCH_NOBLOCK", "1")
        _, rb = _dev_frame(ctx, smo, cur, pv, basic, sigma, p)
        for f in ("active", "nsel", "np0", "nagg", "topk", "gcoords"):
            assert np.array_equal(ra[f], rb[f]), f
    monkeypatch.delenv("NLK_MATCH_NOBLOCK", raising=False)


@pytest.mark.parametrize("psz,size", [(8, (1000, 560)), (10, (1400, 900)), (12, (1600, 1000)), (16, (2000, 1300))])
def test_blocks_of_2x2_targets_equal_target_by_target(ctx, built, synth, monkeypatch, psz, size):
    """With the temporal search radius, patches of 8 x 8 and more put 8 wavefronts on an 8 x 4 tile, each with a block of 2 x 2 targets
    (`k_bm_topk<.., 2>`). Blocks of 4 x 2 (NLK_MATCH_BX2=0) and no blocks at all (NLK_MATCH_NOBLOCK=1) must give the
    same records, for temporal frames with NaN holes and for the smoother."""
    w, h = size
    ch, sigma = 3, 40.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 30 + psz)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1, patch_sz=psz)
    p3 = built.default_params(sigma, built.SMO1, patch_sz=psz)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    holes = prev.copy()
    holes[200:260, 500:640] = np.nan
    holes[:, :3] = np.nan
    for smo, cur, pv, p in [(False, o1, prev, p1), (False, o1, holes, p1), (True, o1, holes, p3)]:
        recs = []
        for env in ({}, {"NLK_MATCH_BX2": "0"}, {"NLK_MATCH_NOBLOCK": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            recs.append(_dev_frame(ctx, smo, cur, pv, None, sigma, p)[1])
            for k in env:
                monkeypatch.delenv(k)
        for other in recs[1:]:
            for f in ("active", "nsel", "np0", "nagg", "topk", "gcoords"):
                assert np.array_equal(recs[0][f], other[f]), (psz, smo, f)


@pytest.mark.parametrize("reach", [1, 2, 3])
def test_mask_replay_on_synthetic_mark_words(ctx, O, monkeypatch, reach):
    """The processed-mask replay alone, on mark words no image produces: dense marks, rows where every
    target marks its right neighbour (runs of ones that fill whole 32-bit words, the carry fix-up of
    k_mask_commit_rows1; chains as long as the row for the iteration of k_mask_commit_rows<2|3>), grid widths
    around the word and lane limits. The default is the row replay on bit planes, NLK_COMMIT_WAVE=1 the
    diagonal replay; both must equal the oracle's serial loop."""
    rng = np.random.default_rng(7 + reach)
    side = 2 * reach + 1
    shapes = [(1, 1), (31, 5), (32, 9), (33, 17), (64, 3), (65, 40), (479, 37), (512, 8), (1000, 13), (2048, 5),
              (2049, 4)]
    for ngx, ngy in shapes:
        for density, right in ((0.5, 0.5), (0.9, 0.97), (0.15, 1.0), (1.0, 1.0)):
            bits = rng.random((ngx * ngy, side * side)) < density
            c = reach * side + reach
            bits[:, c + 1] = rng.random(ngx * ngy) < right   # (di = +1, dj = 0)
            if right == 1.0 and density < 1.0:                # some rows with nothing from above: pure runs
                bits[:, c + 2:] &= (rng.random((ngx * ngy, 1)) < 0.3)
            marks = (bits.astype(np.uint64) << np.arange(side * side, dtype=np.uint64)).sum(axis=1).astype(np.uint64)
            want = O.mask_commit(marks, ngx, ngy, reach)
            d_marks = ctx.upload(marks)
            for wave in (False, True):
                if wave:
                    monkeypatch.setenv("NLK_COMMIT_WAVE", "1")
                else:
                    monkeypatch.delenv("NLK_COMMIT_WAVE", raising=False)
                d_active = ctx.upload(np.full(ngx * ngy, 7, np.uint8))
                ctx.mask_commit(d_marks, ngx, ngy, reach, d_active)
                got = ctx.download(d_active, (ngx * ngy,), np.uint8)
                ctx.free(d_active)
                assert np.array_equal(got, want), (ngx, ngy, density, right, wave, int((got != want).sum()))
            ctx.free(d_marks)
    monkeypatch.delenv("NLK_COMMIT_WAVE", raising=False)


def test_mask_replay_inside_the_group_launch_equals_the_separate_kernels(ctx, built, synth, monkeypatch):
    """Temporal frames of 8 x 8 patches replay the processed mask INSIDE the group kernel's launch (k_group8m:
    workgroup 0 runs the row replay, the others poll generation-tagged decision words); NLK_NO_CHASE=1 runs the
    separate kernels before it. Same decisions for every target, same frame up to the order of the float sums -
    small and full-size grids, one and three channels, NaN holes, the second iteration and the smoother; and a
    context that alternates between the two keeps giving them (the generation of the words moves on)."""
    shapes = [(96, 64, 3, 21), (70, 53, 1, 22), (1000, 560, 3, 23), (1920, 1080, 3, 24)]
    for w, h, ch, seed in shapes:
        sigma = 20.0
        n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, seed)
        o0, o1 = (built.rgb2opp(n0), built.rgb2opp(n1)) if ch == 3 else (n0, n1)
        p1, p2, p3 = (built.default_params(sigma, m) for m in (built.FLT1, built.FLT2, built.SMO1))
        prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        holes = prev.copy()
        holes[h // 4:h // 3, w // 3:w // 2] = np.nan
        holes[:, :2] = np.nan
        calls = [(False, o1, prev, None, p1), (False, o1, holes, None, p1), (False, n1, prev, o1, p2),
                 (True, o1, holes, None, p3)]
        for rep in range(2):
            for smo, cur, pv, basic, p in calls:
                monkeypatch.delenv("NLK_NO_CHASE", raising=False)
                fa, ra = _dev_frame(ctx, smo, cur, pv, basic, sigma, p)
                monkeypatch.setenv("NLK_NO_CHASE", "1")
                fb, rb = _dev_frame(ctx, smo, cur, pv, basic, sigma, p)
                for f in ("active", "nsel", "np0", "nagg", "topk", "gcoords"):
                    assert np.array_equal(ra[f], rb[f]), (w, h, ch, smo, f)
                if p is p1 and pv is prev and min(w, h) > 100:
                    assert 0.05 < 1 - ra["active"].mean() < 0.6   # (the skip really is exercised)
                fa, _ = cases.excuse_flips(fa, fb, cur, f"replay in the launch {w}x{h}x{ch}", most=8)
                cases.assert_close(fa, fb, f"replay in the launch vs separate kernels {w}x{h}x{ch} smoother={smo}")
    monkeypatch.delenv("NLK_NO_CHASE", raising=False)


def test_mask_replay_rescues_itself_when_workgroup_0_never_replays(ctx, built, synth, monkeypatch):
    """The replay inside the group kernel's launch must not depend on the order workgroups are dispatched in (HIP
    promises none): a workgroup whose decision words have not arrived after ~80 us replays the rows it needs itself -
    the same words, the same generation. NLK_CHASE_TEST_SKIP0=1 makes workgroup 0 skip its replay, so EVERY workgroup
    takes that path: same decisions as the separate kernels (which the full-size tests hold against the oracle),
    small grids and 1080p, reach 1 (temporal) and reach 2 (first frame)."""
    for w, h, ch, seed in [(96, 64, 3, 31), (1920, 1080, 3, 32)]:
        sigma = 20.0
        n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, seed)
        o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
        p1 = built.default_params(sigma, built.FLT1)
        monkeypatch.delenv("NLK_CHASE_TEST_SKIP0", raising=False)
        monkeypatch.setenv("NLK_NO_CHASE", "1")
        prev, r0 = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        fb, rb = _dev_frame(ctx, False, o1, prev, None, sigma, p1)
        monkeypatch.delenv("NLK_NO_CHASE", raising=False)
        monkeypatch.setenv("NLK_CHASE_TEST_SKIP0", "1")
        f0, ra0 = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        fa, ra = _dev_frame(ctx, False, o1, prev, None, sigma, p1)
        for f in ("active", "nsel", "np0", "nagg", "topk", "gcoords"):
            assert np.array_equal(ra[f], rb[f]), (w, h, f)
            assert np.array_equal(ra0[f], r0[f]), (w, h, f, "first frame")
        assert 0.05 < 1 - ra["active"].mean() < 0.6
        fa, _ = cases.excuse_flips(fa, fb, o1, f"self-rescued replay {w}x{h}", most=8)
        cases.assert_close(fa, fb, f"self-rescued replay vs separate kernels {w}x{h}")
        f0, _ = cases.excuse_flips(f0, prev, o0, f"self-rescued replay, first frame {w}x{h}", most=8)
        cases.assert_close(f0, prev, f"self-rescued replay vs separate kernels, first frame {w}x{h}")
    monkeypatch.delenv("NLK_CHASE_TEST_SKIP0", raising=False)


def test_group_kernel_dct_forms_agree(ctx, built, synth, monkeypatch):
    """k_group8m runs its DCTs in two forms: the Kronecker form on 16 x 16 x 4 matrix products and the separable
    folded form on 4 x 4 blocks (v_mfma_f32_4x4x1_16B_f32), selectable by pass (NLK_GROUP_SEP: 0 Kronecker in both
    passes, 2 separable pass B, 6 separable in both; the default picks by mode). Same records, same formulas,
    different summation orders: all four frame calls agree to FP noise, one and three channels, NaN holes."""
    for w, h, ch in [(320, 200, 3), (131, 97, 1)]:
        sigma = 20.0
        n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 5)
        o0, o1 = (built.rgb2opp(n0), built.rgb2opp(n1)) if ch == 3 else (n0, n1)
        p1, p2, p3 = (built.default_params(sigma, m) for m in (built.FLT1, built.FLT2, built.SMO1))

        def chain(ref=None):
            f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
            f0i = f0 if ref is None else ref[0]
            hole = f0i.copy()
            hole[h // 5:h // 3, w // 3:w // 2] = np.nan                      # spatial-fallback targets
            f1, _ = _dev_frame(ctx, False, o1, hole, None, sigma, p1)
            f1i = f1 if ref is None else ref[1]
            f2, _ = _dev_frame(ctx, False, o1, hole, f1i, sigma, p2)
            f2i = f2 if ref is None else ref[2]
            s0, _ = _dev_frame(ctx, True, f0i, f2i, None, sigma, p3)
            return f0, f1, f2, s0

        monkeypatch.setenv("NLK_GROUP_SEP", "0")
        a = chain()
        for sep in ("2", "6"):
            monkeypatch.setenv("NLK_GROUP_SEP", sep)
            b = chain(a)
            for name, x, y in zip(("flt1 spatial", "flt1 temporal", "flt2", "smo1"), a, b):
                assert np.isfinite(x).all() and np.isfinite(y).all()
                cases.assert_close(y, x, f"group kernel NLK_GROUP_SEP={sep} vs Kronecker, {name}, {w}x{h}x{ch}")
    monkeypatch.delenv("NLK_GROUP_SEP", raising=False)


def _patch_dist64(img, px, py, q, psz=8):
    """squared patch distance in float64 (mean over the patch and the channels), q = array of packed x | y << 16"""
    a = img[py:py + psz, px:px + psz].astype(np.float64)
    out = np.empty(len(q))
    for i, v in enumerate(q):
        x, y = int(v) & 0xFFFF, int(v) >> 16
        d = img[y:y + psz, x:x + psz].astype(np.float64) - a
        out[i] = (d * d).mean()
    return out


def test_block_summed_match_order(ctx, built, synth, monkeypatch):
    """NLK_MATCH_ORDER=block (opt-in, never the default): the matcher sums a patch distance as four quarter-patch
    sums shared between the targets that hold them (k_match.h: nlk_match_block_sum) instead of in the reference's
    (hy, hx, c) order. The k-NN SETS may then differ from the exact mode's only where two candidates' distances are
    within float-summation noise of each other; whatever path computes a distance - blocks of targets, single
    targets, clipped windows, other tile shapes - gives the same bits; and the filtered frame keeps its PSNR."""
    w, h, ch, sigma = 640, 360, 3, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 11)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1)
    step = p1.patch_sz // 2
    ngx = (w - p1.patch_sz) // step + 1

    def run():
        f0, r0 = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        hole = f0.copy()
        hole[100:140, 200:300] = np.nan                        # spatial-branch targets in the temporal frame (k_bm_wide)
        f1, r1 = _dev_frame(ctx, False, o1, hole, None, sigma, p1)
        return f0, r0, f1, r1, hole

    monkeypatch.delenv("NLK_MATCH_ORDER", raising=False)
    e = run()
    monkeypatch.setenv("NLK_MATCH_ORDER", "block")
    b = run()
    # the same bits from every path in block order: target by target, and the small-grid tiles
    monkeypatch.setenv("NLK_MATCH_NOBLOCK", "1")
    b2 = run()
    monkeypatch.delenv("NLK_MATCH_NOBLOCK", raising=False)
    monkeypatch.setenv("NLK_MTX", "4")
    monkeypatch.setenv("NLK_MTY", "2")
    b3 = run()
    monkeypatch.delenv("NLK_MTX", raising=False)
    monkeypatch.delenv("NLK_MTY", raising=False)
    for other, what in ((b2, "target by target"), (b3, "4 x 2 tiles")):
        for ri in (1, 3):
            for f in ("nsel", "np0", "nagg", "topk", "gcoords"):
                assert np.array_equal(b[ri][f], other[ri][f]), f"block order, {what}: {f} differs from the block path"
    monkeypatch.delenv("NLK_MATCH_ORDER", raising=False)

    worst, ndiff, ntargets = 0.0, 0, 0
    for img, prev_in, re_, rb_ in ((o0, None, e[1], b[1]), (o1, e[4], e[3], b[3])):
        assert np.array_equal(re_["nsel"], rb_["nsel"])
        for t in range(len(re_["nsel"])):
            k = int(re_["nsel"][t])
            se, sb_ = re_["topk"][t, :k], rb_["topk"][t, :k]
            ntargets += 1
            if np.array_equal(np.sort(se), np.sort(sb_)):
                continue
            ndiff += 1
            gy, gx = divmod(t, ngx)
            only = np.concatenate([np.setdiff1d(se, sb_), np.setdiff1d(sb_, se)])
            d_only = _patch_dist64(img, gx * step, gy * step, only)
            d_kth = _patch_dist64(img, gx * step, gy * step, se).max()
            worst = max(worst, float(np.abs(d_only - d_kth).max() / max(d_kth, 1e-30)))
    print(f"block-summed order: {ndiff} of {ntargets} k-NN sets differ; the exchanged candidates lie within "
          f"{worst:.2e} (relative) of the k-th distance")
    assert ndiff <= ntargets // 50
    assert worst <= 2e-5   # a float32 sum of 192 terms: a few ulp
    # the frames: same quality
    for fe, fb, what in ((e[0], b[0], "spatial"), (e[2], b[2], "temporal")):
        assert np.isfinite(fb).all()
        pe, pb = synth.psnr(built.opp2rgb(fe), c1 if what == "temporal" else synth.clean_frame(w, h, ch, 0)), \
            synth.psnr(built.opp2rgb(fb), c1 if what == "temporal" else synth.clean_frame(w, h, ch, 0))
        assert abs(pe - pb) <= 0.02, f"block order, {what}: PSNR {pb:.4f} against {pe:.4f}"


def test_block_summed_match_order_psnr_1080p(ctx, built, synth, monkeypatch):
    """|dPSNR| <= 0.02 dB at C2 (1920x1080x3, sigma 20, FLT1 temporal) between the exact and the block-summed order."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1)
    out = {}
    for order in ("exact", "block"):
        monkeypatch.setenv("NLK_MATCH_ORDER", order)
        f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        f1, rec = _dev_frame(ctx, False, o1, f0, None, sigma, p1)
        out[order] = (synth.psnr(built.opp2rgb(f1), c1), rec)
    monkeypatch.delenv("NLK_MATCH_ORDER", raising=False)
    same = sum(np.array_equal(np.sort(a[:k]), np.sort(b[:k])) for a, b, k in
               zip(out["exact"][1]["topk"][::37], out["block"][1]["topk"][::37], out["exact"][1]["nsel"][::37]))
    print(f"PSNR exact {out['exact'][0]:.4f} dB, block {out['block'][0]:.4f} dB; "
          f"{same} of {len(out['exact'][1]['nsel'][::37])} sampled k-NN sets identical")
    assert abs(out["exact"][0] - out["block"][0]) <= 0.02


def test_smoother_with_a_basic_estimate(ctx, built, O, synth):
    """nlkalman_smooth_frame with bsic1 != NULL (the command line never passes one, the API accepts it,
    src/nlkalman.c:1409): matching and statistics on the basic estimate, the filtered patches from filt1 - and pass B
    of the group kernel on the difference image smoo0 - filt1 laid out with the frame (round 6). One and three
    channels, NaN holes in smoo0; records exact, pixels within the tolerance."""
    for w, h, ch, seed in ((160, 96, 3, 21), (131, 75, 1, 22)):
        sigma = 20.0
        n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, seed)
        o0, o1 = (built.rgb2opp(n0), built.rgb2opp(n1)) if ch == 3 else (n0, n1)
        p1, ps = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.SMO1)
        f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)       # filt1
        f1, _ = _dev_frame(ctx, False, o1, f0, None, sigma, p1)         # smoo0 (no flow: the frames are 2 px apart)
        basic = (0.5 * f0 + 0.5 * o0).astype(np.float32)                # some other estimate of the same frame
        prev = f1.copy()
        prev[h // 4:h // 3, w // 5:w // 2] = np.nan
        prev[:2] = np.nan
        g, rec = _dev_frame(ctx, True, f0, prev, basic, sigma, ps)
        r, tr = O.smooth_frame(f0, prev, basic, sigma, _to_o(O, ps), trace=True)
        _check_records(rec, tr, f"smoother with basic {w}x{h}x{ch}")
        g, _ = cases.excuse_threshold_pixels(g, r, tr, f"smoother with basic {w}x{h}x{ch}", 64)
        cases.assert_close(g, r, f"smoother with basic {w}x{h}x{ch}")


def test_group_kernels_agree_matrix_vs_dpp(ctx, built, synth, monkeypatch):
    """The 8x8 group kernel has two implementations: k_group8m (DCTs on the f32
    matrix cores, the default) and k_group8 (registers + DPP, NLK_GROUP_DPP=1).
    Same records, same formulas, different summation orders: all four frame calls
    must agree to FP noise (and neither is a fallback of the other: both run HIP)."""
    w, h, ch, sigma = 320, 200, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 5)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, p2, p3 = (built.default_params(sigma, m) for m in (built.FLT1, built.FLT2, built.SMO1))

    def chain():
        f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
        hole = f0.copy()
        hole[40:70, 100:160] = np.nan                      # spatial-fallback targets
        f1, _ = _dev_frame(ctx, False, o1, hole, None, sigma, p1)
        f2, _ = _dev_frame(ctx, False, o1, hole, f1, sigma, p2)
        s0, _ = _dev_frame(ctx, True, f0, f2, None, sigma, p3)
        return f0, f1, f2, s0

    monkeypatch.delenv("NLK_GROUP_DPP", raising=False)
    a = chain()
    monkeypatch.setenv("NLK_GROUP_DPP", "1")
    b = chain()
    for name, x, y in zip(("flt1 spatial", "flt1 temporal", "flt2", "smo1"), a, b):
        assert np.isfinite(x).all() and np.isfinite(y).all()
        cases.assert_close(x, y, f"matrix vs DPP group kernel, {name}")


def test_4k_patch12_against_parallel_oracle(ctx, built, O, synth):
    """BASELINE.json configs[2] geometry (3840x2160 RGB, sigma 40, 12x12 patches),
    temporal FLT1. With step 6 > temporal radius 5 a group never reaches another
    grid target, so the processed-mask never fires and the OpenMP oracle is
    deterministic: full-size comparison in seconds."""
    w, h, ch, sigma = 3840, 2160, 3, 40.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 2)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1, patch_sz=12)
    po = _to_o(O, p)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p)
    g, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    assert rec["active"].all()
    r, tr = O.filter_frame(o1, prev, None, sigma, po, nthreads=min(O.max_threads(), 100), trace=True)
    # the integer records of all 229 401 targets of the one 12x12 BASELINE configuration (missing 6):
    # every target is processed here, so the k-NN lists, counts and group members do not depend on the order
    # the oracle's threads took them in
    _check_records(rec, tr, "4K psz12")
    g, _ = cases.excuse_threshold_pixels(g, r, tr, "4K psz12", 256)
    cases.assert_close(g, r, "4K psz12")
    assert abs(synth.psnr(built.opp2rgb(g), c1) - synth.psnr(O.opp2rgb(r), c1)) <= 0.02
    assert synth.psnr(built.opp2rgb(g), c1) > synth.psnr(n1, c1) + 8


def test_smoother_full_size_1080p(ctx, built, O, synth):
    """flt2 -> smo1 at 1080p (BASELINE.json configs[4] last stage), serial oracle."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, ps = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.SMO1)
    f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    f1, _ = _dev_frame(ctx, False, o1, f0, None, sigma, p1)
    g, rec = _dev_frame(ctx, True, f0, f1, None, sigma, ps)
    r, tr = O.smooth_frame(f0, f1, None, sigma, _to_o(O, ps), trace=True)
    _check_records(rec, tr, "smo 1080p")
    g, _ = cases.excuse_threshold_pixels(g, r, tr, "smo 1080p", 64)
    cases.assert_close(g, r, "smo 1080p")


@pytest.mark.parametrize("launch", ["auto", "large-grid", "large-grid-4x2"])
def test_randomised_parameters_and_shapes(ctx, built, O, monkeypatch, launch):
    """120 seeded random configurations: odd image sizes, every supported patch
    size, clipped windows, k larger than the window, group sizes above k, NaN
    holes, second-iteration and smoother calls. Integer records exact, pixels
    within tolerance, for each of them. Small grids launch in a latency-bound shape
    (small tiles, one target at a time); "large-grid" forces the shapes of a full-size
    frame - 8 x 4 match tiles worked in blocks (2 x 2 targets on 8 wavefronts where a full-size frame would use them,
    4 x 2 on 4 wavefronts otherwise; "large-grid-4x2": 4 x 2 everywhere), 4 x 1 group tiles - onto the same images."""
    if launch != "auto":
        for k, v in (("NLK_MATCH_BLOCK", "1"), ("NLK_MTX", "8"), ("NLK_MTY", "4"), ("NLK_GTX", "4")):
            monkeypatch.setenv(k, v)
        if launch == "large-grid-4x2":
            monkeypatch.setenv("NLK_MATCH_BX2", "0")
    # (NLK_RANDOM_SEED / NLK_RANDOM_COUNT run other or longer sequences, e.g. as a soak test)
    rng = np.random.default_rng(int(os.environ.get("NLK_RANDOM_SEED", 2024)))
    want = int(os.environ.get("NLK_RANDOM_COUNT", 120))
    done = 0
    for it in range(5 * want):
        if done == want:
            break
        psz = int(rng.choice([4, 6, 8, 8, 8, 10, 12, 12, 16]))
        step = psz // 2
        ch = int(rng.choice([1, 3]))
        w = int(rng.integers(psz, int(os.environ.get("NLK_RANDOM_MAXW", 90))))
        h = int(rng.integers(psz, int(os.environ.get("NLK_RANDOM_MAXH", 70))))
        smoother = rng.random() < 0.25
        wsz_t = int(rng.integers(1, min(15, 3 * step + step - 1) + 1))
        wsz_x = int(rng.integers(1, min(15, 3 * step + step - 1) + 1))
        npx, npt = int(rng.integers(2, 70)), int(rng.integers(2, 70))
        ntagg = int(rng.integers(1, 45))
        over = dict(patch_sz=psz, search_sz_x=wsz_x, search_sz_t=wsz_t, npatches_x=npx, npatches_t=npt,
                    npatches_tagg=ntagg)
        mode = built.SMO1 if smoother else int(rng.choice([built.FLT1, built.FLT2]))
        sigma = float(rng.choice([10.0, 20.0, 40.0]))
        p = built.default_params(sigma, mode, **over)
        cur = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        kind = rng.integers(0, 3) if not smoother else rng.integers(1, 3)
        prev = None
        if kind >= 1:
            prev = (cur + rng.normal(0, 8, cur.shape)).astype(np.float32)
        if kind == 2:
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            prev[y0:y0 + int(rng.integers(1, 9)), x0:x0 + int(rng.integers(1, 9)), :] = np.nan
            prev[:, :1, :] = np.nan
        basic = None
        if not smoother and mode == built.FLT2:
            basic = (cur + rng.normal(0, 3, cur.shape)).astype(np.float32)
        fn = O.smooth_frame if smoother else O.filter_frame
        r, tr = fn(cur, prev, basic, sigma, _to_o(O, p), trace=True)
        try:
            g, rec = _dev_frame(ctx, smoother, cur, prev, basic, sigma, p)
        except built.NlkError as e:  # combinations the kernels reject loudly (reach > 3)
            assert "reach" in str(e) or "LDS" in str(e), str(e)
            continue
        what = f"random #{it}: {w}x{h}x{ch} psz{psz} sx{wsz_x} st{wsz_t} nx{npx} nt{npt} na{ntagg} " \
               f"mode{mode} prev{kind} sigma{sigma}"
        _check_records(rec, tr, what)
        # a pixel whose summed weight sits AT the reference's absolute threshold (aggr > 1e-6 ?
        # normalise : pass the input through, src/nlkalman.c:939-942) may fall on either side with
        # the weights summed in another order: those pixels are excused, and only those
        # (within 1e-4 relative of it: 100x the spread of a float sum of ~100 weights)
        edge = np.abs(tr["aggr"] - 1e-6) <= 1e-10
        assert edge.sum() <= max(4, 0.02 * edge.size), what
        g = np.where(edge[..., None], r, g)
        cases.assert_close(g, r, what, maxabs=5e-3, rmse=5e-4)
        done += 1
    assert done == want


# ---------------------------------------------------------------- round 2: the configs of BASELINE.json that
# round 1 left untested at their real sizes (VERDICT r1 "next" #1)

def test_config1_256x256_gray_against_serial_oracle(ctx, built, O, synth):
    """BASELINE.json configs[0]: one 256x256 grayscale frame + previous frame, sigma 20, 8x8
    patches, against the serial oracle (the reference's OpenMP-off order): FLT1 spatial (frame 0),
    FLT1 temporal and FLT2 temporal (frame 1). Records exact, pixels within tolerance."""
    w, h, ch, sigma = 256, 256, 1, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 0)
    p1, p2 = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.FLT2)
    r0, tr0 = O.filter_frame(n0, None, None, sigma, _to_o(O, p1), trace=True)
    g0, rec0 = _dev_frame(ctx, False, n0, None, None, sigma, p1)
    _check_records(rec0, tr0, "C1 flt1 spatial")
    cases.assert_close(g0, r0, "C1 flt1 spatial")
    r1, tr1 = O.filter_frame(n1, r0, None, sigma, _to_o(O, p1), trace=True)
    g1, rec1 = _dev_frame(ctx, False, n1, r0, None, sigma, p1)
    _check_records(rec1, tr1, "C1 flt1 temporal")
    cases.assert_close(g1, r1, "C1 flt1 temporal")
    assert 0.2 < 1 - tr1["active"].mean() < 0.5
    r2, tr2 = O.filter_frame(n1, r0, r1, sigma, _to_o(O, p2), trace=True)
    g2, rec2 = _dev_frame(ctx, False, n1, r0, r1, sigma, p2)
    _check_records(rec2, tr2, "C1 flt2 temporal")
    cases.assert_close(g2, r2, "C1 flt2 temporal")
    assert abs(synth.psnr(g2, c1) - synth.psnr(r2, c1)) <= 0.02
    assert synth.psnr(g2, c1) > synth.psnr(n1, c1) + 8


def test_second_iteration_full_size_1080p(ctx, built, O, synth):
    """The middle stage of BASELINE.json configs[4] at full size: FLT2 temporal at 1920x1080 RGB
    (bsic1 = the FLT1 output, deno0 = previous FLT2 output stood in for by the previous FLT1
    output), serial oracle."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, p2 = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.FLT2)
    f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    f1, _ = _dev_frame(ctx, False, o1, f0, None, sigma, p1)
    g, rec = _dev_frame(ctx, False, o1, f0, f1, sigma, p2)
    r, tr = O.filter_frame(o1, f0, f1, sigma, _to_o(O, p2), trace=True)
    assert tr["active"].all()                       # npatches_tagg = 1: the mask skip never fires
    _check_records(rec, tr, "flt2 1080p")
    g, _ = cases.excuse_threshold_pixels(g, r, tr, "flt2 1080p", 64)
    cases.assert_close(g, r, "flt2 1080p")
    assert abs(synth.psnr(built.opp2rgb(g), c1) - synth.psnr(O.opp2rgb(r), c1)) <= 0.02


def test_config4_decomposition_1080p_eight_strips(ctx, built, synth):
    """BASELINE.json configs[3]'s decomposition — 1920x1080 RGB split into 8 row strips, exact
    mode (match per strip -> concatenated mark words -> whole-grid mask replay -> group per
    strip -> accumulator halos added) — run sequentially on one GPU: active flags identical to
    the whole-frame call, pixels equal up to the summation order of the accumulator."""
    import importlib
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    w, h, ch, sigma, world = 1920, 1080, 3, 20.0, 8
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p)
    whole, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    plan = strips.strip_plan(h, p.patch_sz, max(p.search_sz_x, p.search_sz_t), world)
    assert len(plan) == world
    step = p.patch_sz // 2
    ngx, ngy = (w - p.patch_sz) // step + 1, (h - p.patch_sz) // step + 1
    assert sum(s["gy1"] - s["gy0"] for s in plan) == ngy
    d_marks = ctx.upload(np.zeros(ngx * ngy, np.uint64))
    d_active = ctx.upload(np.zeros(ngx * ngy, np.uint8))
    bufs = []
    for s in plan:
        cur_s = np.ascontiguousarray(o1[s["Y0"]:s["Y1"]])
        prev_s = np.ascontiguousarray(prev[s["Y0"]:s["Y1"]])
        d_cur, d_prev = ctx.upload(cur_s), ctx.upload(prev_s)
        oy, ngy_l = s["gy0"] * step - s["Y0"], s["gy1"] - s["gy0"]
        reach = ctx.strip_match(d_marks + 8 * s["gy0"] * ngx, d_cur, d_prev, None, w, cur_s.shape[0], ch,
                                sigma, p, oy, ngy_l)
        bufs.append((d_cur, d_prev, cur_s.shape[0], oy, ngy_l))
    ctx.mask_commit(d_marks, ngx, ngy, reach, d_active)
    assert np.array_equal(ctx.download(d_active, (ngx * ngy,), np.uint8), rec["active"])
    acc = np.zeros((ch + 1, h, w), np.float32)
    for s, (d_cur, d_prev, hl, oy, ngy_l) in zip(plan, bufs):
        ctx.strip_match(None, d_cur, d_prev, None, w, hl, ch, sigma, p, oy, ngy_l)
        d_acc = ctx.upload(np.zeros((ch + 1, hl, w), np.float32))
        ctx.strip_group(d_acc, d_active + s["gy0"] * ngx)
        acc[:, s["Y0"]:s["Y1"]] += ctx.download(d_acc, (ch + 1, hl, w))
        for x in (d_cur, d_prev, d_acc):
            ctx.free(x)
    d_acc, d_cur, d_out = ctx.upload(acc), ctx.upload(o1), ctx.alloc(o1.nbytes)
    ctx.frame_normalize(d_out, d_acc, d_cur, w, h, ch, 0, h)
    got = ctx.download(d_out, o1.shape)
    for x in (d_acc, d_cur, d_out, d_marks, d_active):
        ctx.free(x)
    got, _ = cases.excuse_flips(got, whole, o1, "8 exact strips at 1080p vs whole frame", 64)
    cases.assert_close(got, whole, "8 exact strips at 1080p vs whole frame", maxabs=5e-4, rmse=5e-5)


def test_accumulators_of_filter_and_smoother_interleaved(ctx, built, O, synth):
    """The accumulate / normalise split of the C-ABI (include/nlk_hip.h): a planar accumulator holds ch weighted
    sums and the weights, whatever kernel filled it - no state in the context says how to read it (ADVICE r3:
    the smoother on the matrix cores used to leave (member - image) sums and a hidden flag for the normaliser).
    A filter accumulate, then a smoother accumulate on the SAME context, then the two normalisations in that
    order: each must equal its whole-frame call, and the oracle's."""
    w, h, ch, sigma = 200, 136, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 23)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1, ps = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.SMO1)
    f0, _ = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    f1, _ = _dev_frame(ctx, False, o1, f0, None, sigma, p1)
    whole_f, _ = _dev_frame(ctx, False, o1, f0, None, sigma, p1)
    whole_s, _ = _dev_frame(ctx, True, f0, f1, None, sigma, ps)
    ngy = (h - 8) // 4 + 1
    d_o1, d_f0, d_f1 = ctx.upload(o1), ctx.upload(f0), ctx.upload(f1)
    acc_f = ctx.upload(np.zeros((ch + 1, h, w), np.float32))
    acc_s = ctx.upload(np.zeros((ch + 1, h, w), np.float32))
    out_f, out_s = ctx.alloc(o1.nbytes), ctx.alloc(o1.nbytes)
    ctx.frame_accumulate(acc_f, d_o1, d_f0, None, w, h, ch, sigma, p1, 0, ngy)                  # filter
    ctx.frame_accumulate(acc_s, d_f0, d_f1, None, w, h, ch, sigma, ps, 0, ngy, smoother=True)   # smoother, same ctx
    ctx.frame_normalize(out_f, acc_f, d_o1, w, h, ch, 0, h)
    ctx.frame_normalize(out_s, acc_s, d_f0, w, h, ch, 0, h)
    got_f, got_s = ctx.download(out_f, o1.shape), ctx.download(out_s, o1.shape)
    # the accumulator means what the header says: ch weighted sums of member pixels, then the weights
    a = ctx.download(acc_s, (ch + 1, h, w))
    ok = a[ch] > 1e-6
    assert np.abs(np.moveaxis(a[:ch], 0, -1)[ok] / a[ch][ok][:, None] - got_s[ok]).max() <= 1e-3
    for x in (d_o1, d_f0, d_f1, acc_f, acc_s, out_f, out_s):
        ctx.free(x)
    cases.assert_close(got_f, whole_f, "filter accumulate / normalise around a smoother accumulate", maxabs=5e-4, rmse=5e-5)
    cases.assert_close(got_s, whole_s, "smoother accumulate / normalise after a filter accumulate", maxabs=5e-4, rmse=5e-5)
    cases.assert_close(got_s, O.smooth_frame(f0, f1, None, sigma, _to_o(O, ps)), "smoother through the split API vs oracle")


def test_local_mode_single_patch(ctx, built, O):
    """SURVEY.md A9 (reference: src/nlkalman.c:815-857; smoother :1699-1730, :1795-1804):
    npatches_x <= 1 / npatches_t <= 1 put the targets concerned into the single-patch "local"
    mode — the filter aggregates nothing for them (np0 = np1 = 0), the smoother passes the target
    patch through. Every combination, with NaN holes so that both kinds of target occur."""
    rng = np.random.default_rng(9)
    w, h = 52, 44
    for ch in (1, 3):
        cur = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        prev = (cur + rng.normal(0, 6, cur.shape)).astype(np.float32)
        prev[10:19, 20:31] = np.nan
        prev[:, :1] = np.nan
        basic = (cur + rng.normal(0, 3, cur.shape)).astype(np.float32)
        for over in (dict(npatches_x=1), dict(npatches_t=1), dict(npatches_x=1, npatches_t=1),
                     dict(npatches_x=0, npatches_t=0)):
            for mode, pv, bs in ((built.FLT1, None, None), (built.FLT1, prev, None), (built.FLT2, prev, basic)):
                p = built.default_params(20.0, mode, **over)
                r, tr = O.filter_frame(cur, pv, bs, 20.0, _to_o(O, p), trace=True)
                g, rec = _dev_frame(ctx, False, cur, pv, bs, 20.0, p)
                what = f"local filter ch{ch} {over} mode{mode} prev{pv is not None}"
                _check_records(rec, tr, what)
                cases.assert_close(g, r, what)
                if over.get("npatches_x", 2) <= 1 and pv is None:
                    assert np.array_equal(g, cur), what     # nothing aggregated: the input passes through
            ps = built.default_params(20.0, built.SMO1, **{k: v for k, v in over.items() if k == "npatches_t"})
            r, tr = O.smooth_frame(cur, prev, None, 20.0, _to_o(O, ps), trace=True)
            g, rec = _dev_frame(ctx, True, cur, prev, None, 20.0, ps)
            _check_records(rec, tr, f"local smoother ch{ch} {over}")
            cases.assert_close(g, r, f"local smoother ch{ch} {over}")


# ---------------------------------------------------------------- deterministic aggregation / bands

def _frame(ctx, args, sigma, p, smoother=False):
    out, _ = _dev_frame(ctx, smoother, *args, sigma, p)
    return out


@pytest.mark.parametrize("psz", [8, 12, 6])
def test_deterministic_aggregation_is_bit_reproducible(built, O, synth, psz):
    """nlk_ctx_set_deterministic: the accumulator tiles go to slabs and are summed in a fixed order
    (k_gather.h) instead of being added with float atomics. Two runs must then be bit-identical
    (filter with and without a previous frame, NaN holes, second iteration, smoother), equal the
    default mode to summation-order noise, and match the oracle like it does."""
    w, h, ch, sigma = 200, 136, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 17)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p1 = built.default_params(sigma, built.FLT1, patch_sz=psz, search_sz_x=min(10, 3 * (psz // 2)))
    p2 = built.default_params(sigma, built.FLT2, patch_sz=psz, search_sz_x=min(10, 3 * (psz // 2)))
    p3 = built.default_params(sigma, built.SMO1, patch_sz=psz)
    det, dflt = built.Context(0), built.Context(0)
    det.set_deterministic(True)
    try:
        def chain(c):
            f0 = _frame(c, (o0, None, None), sigma, p1)
            hole = f0.copy()
            hole[40:70, 100:160] = np.nan                  # spatial-branch groups inside a temporal frame
            f1 = _frame(c, (o1, hole, None), sigma, p1)
            f2 = _frame(c, (o1, hole, f1), sigma, p2)
            s0 = _frame(c, (f0, f2, None), sigma, p3, smoother=True)
            return f0, f1, f2, s0
        a, b = chain(det), chain(det)
        # the default mode stage by stage on the deterministic run's outputs (a free-running second chain would
        # feed its second iteration a basic estimate that differs by summation noise: a near-tied k-NN rank
        # flips and moves a few dozen samples by hundredths - not what this test is about)
        hole_a = a[0].copy()
        hole_a[40:70, 100:160] = np.nan
        d = (_frame(dflt, (o0, None, None), sigma, p1), _frame(dflt, (o1, hole_a, None), sigma, p1),
             _frame(dflt, (o1, hole_a, a[1]), sigma, p2), _frame(dflt, (a[0], a[2], None), sigma, p3, smoother=True))
        for name, x, y, z in zip(("flt1 spatial", "flt1 temporal", "flt2", "smo1"), a, b, d):
            assert np.array_equal(x, y, equal_nan=True), f"psz {psz} {name}: two deterministic runs differ"
            # (default mode: float atomics in varying order; a pixel whose summed weight sits at the
            # reference's 1e-6 threshold can fall on either side of it from run to run: a few dozen
            # samples of the second iteration's single-member groups with the small patches)
            cur = o0 if name == "flt1 spatial" else (x * 0 + a[0] if name == "smo1" else o1)   # (the pass-through image)
            x, _ = cases.excuse_flips(x, z, cur, f"psz {psz} {name}: deterministic vs default mode", 150)
            cases.assert_close(x, z, f"psz {psz} {name}: deterministic vs default mode")
        po = _to_o(O, p1)
        r0, tr0 = O.filter_frame(o0, None, None, sigma, po, trace=True)
        g0, _ = cases.excuse_threshold_pixels(a[0], r0, tr0, f"psz {psz}: deterministic vs oracle", 16)
        cases.assert_close(g0, r0, f"psz {psz}: deterministic vs oracle")
    finally:
        det.close()
        dflt.close()


def test_deterministic_full_size_1080p_runs_are_identical(built, synth):
    """Run-to-run equality at BASELINE.json configs[1] size, the property the atomics cannot give."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 1)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    det, dflt = built.Context(0), built.Context(0)
    det.set_deterministic(True)
    try:
        prev = _frame(det, (o0, None, None), sigma, p)
        a = _frame(det, (o1, prev, None), sigma, p)
        b = _frame(det, (o1, prev, None), sigma, p)
        assert np.array_equal(a, b)
        d1 = _frame(dflt, (o1, prev, None), sigma, p)
        a, _ = cases.excuse_flips(a, d1, o1, "deterministic vs atomics at 1080p", 64)
        cases.assert_close(a, d1, "deterministic vs atomics at 1080p")
    finally:
        det.close()
        dflt.close()


@pytest.mark.parametrize("bands", [2, 3])
def test_banded_two_stream_pipeline_equals_single_stream(built, synth, monkeypatch, bands):
    """NLK_BANDS: the grid rows of a frame in bands on two streams (match / mask replay / filtering of
    consecutive bands overlap). Same mask decisions, same records, same pixels up to the order of
    the accumulator's atomic adds."""
    w, h, ch, sigma = 320, 400, 3, 20.0
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 23)
    o0, o1 = built.rgb2opp(n0), built.rgb2opp(n1)
    p = built.default_params(sigma, built.FLT1)
    c = built.Context(0)
    try:
        monkeypatch.delenv("NLK_BANDS", raising=False)
        f0, r0 = _dev_frame(c, False, o0, None, None, sigma, p)
        f1, r1 = _dev_frame(c, False, o1, f0, None, sigma, p)
        monkeypatch.setenv("NLK_BANDS", str(bands))
        g0, q0 = _dev_frame(c, False, o0, None, None, sigma, p)
        g1, q1 = _dev_frame(c, False, o1, f0, None, sigma, p)
        for a, b in ((r0, q0), (r1, q1)):
            for key in ("active", "nsel", "np0", "nagg", "topk", "gcoords"):
                assert np.array_equal(a[key], b[key]), key
        assert 0.2 < 1 - r1["active"].mean() < 0.5
        cases.assert_close(g0, f0, f"{bands} bands vs one, spatial", maxabs=5e-4, rmse=5e-5)
        cases.assert_close(g1, f1, f"{bands} bands vs one, temporal", maxabs=5e-4, rmse=5e-5)
    finally:
        c.close()


# ---------------------------------------------------------------- the rest of the reference's parameter space

def test_odd_patch_sizes_other_channel_counts_and_long_reaches(ctx, built, O):
    """What round 1 rejected (VERDICT r1 "missing" 5): the reference takes any patch size
    (src/nlkalman.c:524-525), any channel count (:555-560) and any search radius (:637-639).
    Odd patch sizes and 2 / 4 channels run on k_bm_generic + k_groupp; a group reach of more than 3
    grid cells (e.g. patch 4 with the default radius 10, or --f1_st 20) replays the mask from the
    coordinate lists; a radius above 15 (more than 1024 candidates) searches with k_bm_generic.
    Records exact, pixels within tolerance, for filter (spatial, temporal with NaN holes, second
    iteration) and smoother."""
    rng = np.random.default_rng(77)
    configs = [
        dict(psz=7, ch=3, w=61, h=47, over={}),
        dict(psz=5, ch=1, w=40, h=52, over=dict(search_sz_x=6, search_sz_t=3)),
        dict(psz=9, ch=2, w=57, h=44, over={}),
        dict(psz=8, ch=4, w=48, h=40, over={}),
        dict(psz=11, ch=3, w=50, h=46, over=dict(npatches_t=12)),
        dict(psz=13, ch=1, w=70, h=48, over={}),
        dict(psz=15, ch=3, w=64, h=50, over=dict(npatches_x=20, npatches_t=20)),
        dict(psz=3, ch=3, w=33, h=30, over=dict(search_sz_x=3, search_sz_t=2)),
        dict(psz=4, ch=3, w=44, h=36, over={}),                                 # reach 5 (radius 10 / step 2)
        dict(psz=8, ch=3, w=80, h=64, over=dict(search_sz_t=20, search_sz_x=4)),  # reach 5, 1681 candidates
        dict(psz=8, ch=1, w=90, h=70, over=dict(search_sz_x=17)),                # 1225 candidates
        dict(psz=6, ch=3, w=50, h=44, over=dict(search_sz_x=12, search_sz_t=12)),  # reach 4
    ]
    for cfg in configs:
        psz, ch, w, h = cfg["psz"], cfg["ch"], cfg["w"], cfg["h"]
        cur = rng.uniform(0, 255, (h, w, ch)).astype(np.float32)
        prev = (cur + rng.normal(0, 8, cur.shape)).astype(np.float32)
        prev[h // 3:h // 3 + 5, w // 2:w // 2 + 6] = np.nan
        prev[:, :1] = np.nan
        basic = (cur + rng.normal(0, 3, cur.shape)).astype(np.float32)
        for mode, pv, bs, smo in ((built.FLT1, None, None, False), (built.FLT1, prev, None, False),
                                  (built.FLT2, prev, basic, False), (built.SMO1, prev, None, True)):
            over = dict(cfg["over"])
            if smo:
                over.pop("search_sz_x", None)
                over.pop("npatches_x", None)
            p = built.default_params(20.0, mode, patch_sz=psz, **over)
            fn = O.smooth_frame if smo else O.filter_frame
            r, tr = fn(cur, pv, bs, 20.0, _to_o(O, p), trace=True)
            g, rec = _dev_frame(ctx, smo, cur, pv, bs, 20.0, p)
            what = f"psz {psz} ch {ch} {w}x{h} {cfg['over']} mode {mode} prev {pv is not None}"
            _check_records(rec, tr, what)
            edge = np.abs(tr["aggr"] - 1e-6) <= 1e-10   # (see test_randomised_parameters_and_shapes)
            g = np.where(edge[..., None], r, g)
            cases.assert_close(g, r, what, maxabs=5e-3, rmse=5e-4)


_MULTIDEV_SCRIPT = r"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = 320, 256, 3, 20.0
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 5)
o0, o1 = pkg.rgb2opp(n0), pkg.rgb2opp(n1)
p1, p2, p3 = (pkg.default_params(sigma, m) for m in (pkg.FLT1, pkg.FLT2, pkg.SMO1))
# every stage is fed the FIRST run's outputs of the stages before it (a 1e-4 difference in an input
# can flip a near-tied k-NN rank and move a few output samples by tenths)
ref = dict(np.load(sys.argv[3])) if len(sys.argv) > 3 else {}
f0 = pkg.filter_frame(o0, None, None, sigma, p1)
hole = ref.get("f0", f0).copy(); hole[100:130, 50:90] = np.nan
f1 = pkg.filter_frame(o1, hole, None, sigma, p1)
f2 = pkg.filter_frame(o1, hole, ref.get("f1", f1), sigma, p2)
s0 = pkg.smooth_frame(ref.get("f0", f0), ref.get("f2", f2), None, sigma, p3)
np.savez(sys.argv[2], f0=f0, f1=f1, f2=f2, s0=s0, o0=o0, o1=o1)
"""


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0", "0,1"])
def test_c_api_split_over_devices(built, tmp_path, devices):
    """NLK_DEVICES: the drop-in C API (libnlkalman.so, host/multidev.c) cuts a frame call into row
    strips over the listed devices - match per strip, mark words to every device, whole-grid mask
    replay, group per strip, accumulator halos device to device - all in C. The one-GPU test box lists
    device 0 several times (separate contexts, peer copies onto the same device): the result must
    equal the single-device call up to the order of the accumulator's atomic adds."""
    import subprocess
    import sys
    if devices == "0,1" and built.hip().nlk_device_count() < 2:
        pytest.skip("two distinct devices: needs a multi-GPU box (every entry point selects its context's device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(_MULTIDEV_SCRIPT)
    outs = {}
    for tag, env in (("one", {}), ("split", {"NLK_DEVICES": devices})):
        e = dict(os.environ, **env)
        e.pop("NLK_DEVICES", None) if tag == "one" else None
        args = [sys.executable, str(script), root, str(tmp_path / f"{tag}.npz")]
        if tag == "split":
            args.append(str(tmp_path / "one.npz"))
        try:
            r = subprocess.run(args, env=e, capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired as ex:   # (a first run on two real devices must report, not hang the suite)
            pytest.fail(f"NLK_DEVICES={devices}: the {tag} run did not finish in 240 s and was killed; its stderr so far:\n"
                        + ((ex.stderr or b"").decode(errors="replace") if isinstance(ex.stderr, bytes) else (ex.stderr or ""))[-2000:])
        assert r.returncode == 0, r.stderr[-2000:]
        with np.load(tmp_path / f"{tag}.npz") as z:
            outs[tag] = {k: z[k] for k in z.files}
    for k, cur in (("f0", "o0"), ("f1", "o1"), ("f2", "o1"), ("s0", "f0")):
        got, _ = cases.excuse_flips(outs["split"][k], outs["one"][k], outs["one"][cur], f"NLK_DEVICES={devices}: {k}", 16)
        cases.assert_close(got, outs["one"][k], f"NLK_DEVICES={devices}: {k}", maxabs=5e-4, rmse=5e-5)


@pytest.mark.parametrize("who", ["0", "1"])
def test_c_api_device_error_exits_instead_of_hanging(built, tmp_path, who):
    """An error on any device thread of the NLK_DEVICES path (the caller's own = device 0, or a worker) must end
    the process with status 1 like the reference's `exit(1)` (src/nlkalman.c:165-177) - not hang in exit
    handlers that join threads parked at a barrier. NLK_MULTI_TEST_FAIL injects the error."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fail.py"
    script.write_text('''
import importlib, sys
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
n0, _, _ = synth.noisy_pair(320, 256, 3, 20.0, 3)
pkg.filter_frame(pkg.rgb2opp(n0), None, None, 20.0, pkg.default_params(20.0, pkg.FLT1, search_sz_x=5))
print("survived")
''')
    e = dict(os.environ, NLK_DEVICES="0,0", NLK_MULTI_TEST_FAIL=who)
    r = subprocess.run([sys.executable, str(script), root], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stderr[-1000:])
    assert "injected failure" in r.stderr and "survived" not in r.stdout


def test_c_api_device_list_falls_back_to_one_device(built, tmp_path):
    """NLK_DEVICES with a configuration that cannot be split (a group reaching more than 3 grid cells:
    4 x 4 patches with the default spatial radius 10) or with one listed index must behave like the
    single-device call, not exit (host/multidev.c)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run4.py"
    script.write_text('''
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
n0, _, _ = synth.noisy_pair(120, 96, 3, 20.0, 3)
p = pkg.default_params(20.0, pkg.FLT1, patch_sz=4)
np.save(sys.argv[2], pkg.filter_frame(pkg.rgb2opp(n0), None, None, 20.0, p))
''')
    outs = {}
    for tag, env in (("one", {}), ("list2", {"NLK_DEVICES": "0,0"}), ("list1", {"NLK_DEVICES": "0"})):
        e = {k: v for k, v in os.environ.items() if k != "NLK_DEVICES"}
        e.update(env)
        out = tmp_path / f"{tag}.npy"
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        if tag == "list2":
            assert "not split" in r.stderr
        outs[tag] = np.load(out)
    import importlib
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    cur = built.rgb2opp(synth.noisy_pair(120, 96, 3, 20.0, 3)[0])
    for tag in ("list2", "list1"):
        got, _ = cases.excuse_flips(outs[tag], outs["one"], cur, f"NLK_DEVICES fallback ({tag})", 16)
        cases.assert_close(got, outs["one"], f"NLK_DEVICES fallback ({tag})", maxabs=5e-4, rmse=5e-5)
