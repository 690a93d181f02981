"""The N-strip step driven from C (include/nlk_hip.h: nlk_strips_*, csrc/strips.hip) on the one GPU of the test
box: every strip of the decomposition in this process on device 0 (device copies between the strips), and one
rank on RCCL (communicator of one: the library loads, the grouped calls run, the step is captured into a graph).
Reference analogue: the row split of src/nlkalman.c:586; what must hold: the decisions of the whole-frame call
(= the serial order) and its pixels up to the order of the accumulator's adds."""
import numpy as np
import pytest

import cases
from test_gpu_parity import _dev_frame

pytestmark = pytest.mark.gpu


def _frames(built, synth, w, h, ch, sigma, seed):
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, seed)
    return built.rgb2opp(n0), built.rgb2opp(n1)


def _run_strips(built, ctx, world, o1, prev, sigma, p, smoother=False, overlap=True, rccl=False, graph=False, steps=2):
    h, w, ch = o1.shape
    S = built.Strips([0] * (1 if rccl else world), 0, world, w, h, ch, sigma, p, smoother=smoother, have_prev=prev is not None)
    try:
        if rccl:
            S.rccl_init(built.Strips.unique_id())
        S.set_options(overlap=overlap, graph=graph)
        d_cur, d_prev = ctx.upload(o1), (ctx.upload(prev) if prev is not None else None)
        for i in range(S.nlocal):
            S.load(i, d_cur, d_prev)
        for _ in range(steps):          # (a second step over the same buffers: nothing left over from the first)
            S.step()
        S.sync()
        out = np.zeros_like(o1)
        for i in range(S.nlocal):
            y0, y1, rows = S.download_rows(i)
            out[y0:y1] = rows
        _, _, _, act = S.own_rows(S.nlocal - 1)   # (the last strip's decisions cover the whole grid)
        g = S.geometry(0)
        ngx, ngy = (w - p.patch_sz) // (p.patch_sz // 2) + 1, (h - p.patch_sz) // (p.patch_sz // 2) + 1
        active = ctx.download(act, (ngx * ngy,), np.uint8)
        info = (S.transport(), S.stats())
        for d in (d_cur, d_prev):
            if d:
                ctx.free(d)
        return out, active, info
    finally:
        S.close()


@pytest.mark.parametrize("world", [2, 5, 8])
def test_every_strip_in_one_process_equals_the_whole_frame_1080p(ctx, built, synth, world):
    """BASELINE.json configs[3]'s decomposition at 1920x1080 RGB, FLT1 temporal, 2 / 5 / 8 strips on device 0:
    decisions identical to the whole-frame call, pixels equal up to summation order; with and without matching
    the interior rows while the halo travels."""
    w, h, ch, sigma = 1920, 1080, 3, 20.0
    o0, o1 = _frames(built, synth, w, h, ch, sigma, 1)
    p = built.default_params(sigma, built.FLT1)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p)
    whole, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    for overlap in (True, False):
        got, active, info = _run_strips(built, ctx, world, o1, prev, sigma, p, overlap=overlap)
        assert np.array_equal(active, rec["active"]), f"{world} strips (overlap {overlap}): decisions differ"
        got, _ = cases.excuse_flips(got, whole, o1, f"{world} strips in C (overlap {overlap})", 64)
        cases.assert_close(got, whole, f"{world} strips in C (overlap {overlap})", maxabs=5e-4, rmse=5e-5)
    assert "device copies" in info[0]


def test_strips_in_c_first_frame_second_iteration_free_and_smoother(ctx, built, synth):
    """The other kinds of call through the same step: a first frame (no previous frame: nothing to exchange but
    the mark words and the accumulator halos; mask reach 2), and the smoother with a previous frame that has NaN
    holes on a seam between two strips."""
    w, h, ch, sigma = 640, 480, 3, 20.0
    o0, o1 = _frames(built, synth, w, h, ch, sigma, 4)
    p1, ps = built.default_params(sigma, built.FLT1), built.default_params(sigma, built.SMO1)
    whole0, rec0 = _dev_frame(ctx, False, o0, None, None, sigma, p1)
    got0, act0, _ = _run_strips(built, ctx, 3, o0, None, sigma, p1)
    assert np.array_equal(act0, rec0["active"])
    got0, _ = cases.excuse_flips(got0, whole0, o0, "first frame over 3 strips in C", 64)
    cases.assert_close(got0, whole0, "first frame over 3 strips in C", maxabs=5e-4, rmse=5e-5)
    f1, _ = _dev_frame(ctx, False, o1, whole0, None, sigma, p1)
    f1[150:170, 200:300] = np.nan          # (480 rows over 4 strips: seams near rows 120, 240, 360)
    f1[236:246, 400:420] = np.nan
    wholes, recs = _dev_frame(ctx, True, whole0, f1, None, sigma, ps)
    gots, acts, _ = _run_strips(built, ctx, 4, whole0, f1, sigma, ps, smoother=True)
    assert np.array_equal(acts, recs["active"])
    gots, _ = cases.excuse_flips(gots, wholes, whole0, "smoother over 4 strips in C", 64)
    cases.assert_close(gots, wholes, "smoother over 4 strips in C", maxabs=5e-4, rmse=5e-5)


@pytest.mark.parametrize("graph", [False, True])
def test_one_rank_on_rccl(ctx, built, synth, graph):
    """The RCCL transport with a communicator of one rank (all a one-GPU box can hold: RCCL refuses two ranks on
    one device): librccl is found and loaded, the communicator is made from a unique id, the grouped calls of the
    step run (no neighbours: only the mark-word group is skipped at world 1) - and with `graph` the step is
    captured into a HIP graph and replayed, or falls back to plain launches by itself."""
    w, h, ch, sigma = 640, 480, 3, 20.0
    o0, o1 = _frames(built, synth, w, h, ch, sigma, 6)
    p = built.default_params(sigma, built.FLT1)
    prev, _ = _dev_frame(ctx, False, o0, None, None, sigma, p)
    whole, rec = _dev_frame(ctx, False, o1, prev, None, sigma, p)
    got, active, (transport, (phases, issue_us, replayed)) = _run_strips(built, ctx, 1, o1, prev, sigma, p, rccl=True,
                                                                          graph=graph, steps=4)
    assert "rccl" in transport
    assert np.array_equal(active, rec["active"])
    got, _ = cases.excuse_flips(got, whole, o1, "one rank on RCCL", 64)
    cases.assert_close(got, whole, "one rank on RCCL", maxabs=5e-4, rmse=5e-5)
    print(f"one rank on RCCL (graph requested {graph}, replayed {replayed}): {issue_us:.1f} us to enqueue a step; {transport}")
