"""CPU tests of the N > 1 path: strip plan + neighbour exchanges over gloo
(world_size 2 and 3), with the oracle's row-strip form standing in for the HIP
kernels (same accumulate / normalize contract as include/nlk_hip.h)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, CH, SIGMA, SEED = 64, 96, 3, 20.0, 21


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _frames(O, synth):
    n0, n1, _ = synth.noisy_pair(W, H, CH, SIGMA, SEED)
    o0, o1 = O.rgb2opp(n0), O.rgb2opp(n1)
    p = O.default_params(SIGMA, O.FLT1)
    prev = O.filter_frame(o0, None, None, SIGMA, p)
    return o1, prev, p


def _callbacks(O, p):
    def accumulate(acc, cur, prev, oy, ngy):
        a = acc.numpy()
        O.frame_accumulate(a, cur.numpy(), prev.numpy(), None, SIGMA, p, oy, ngy)

    def normalize(out, acc, cur, y0, y1):
        out[y0:y1] = torch.from_numpy(O.frame_normalize(acc.numpy(), cur.numpy(), y0, y1))[y0:y1]
    return accumulate, normalize


def _phases(O, p, state):
    """Oracle-backed (match, commit, group) callbacks of the exact mode."""
    def match(marks, cur, prev, oy, ngy):
        m = np.zeros(marks.numel(), np.uint64)
        state.update(cur=cur.numpy(), prev=prev.numpy(), oy=oy, ngy=ngy)
        r = O.strip_match(m, state["cur"], state["prev"], None, SIGMA, p, oy, ngy)
        marks.copy_(torch.from_numpy(m.view(np.int64)))
        return r

    def commit(marks_full, ngx, ngy, reach, active_full):
        act = O.mask_commit(marks_full.numpy().view(np.uint64), ngx, ngy, reach)
        active_full.copy_(torch.from_numpy(act))

    def group(acc, active):
        O.strip_group(acc.numpy(), active.numpy(), state["cur"], state["prev"], None, SIGMA, p,
                      state["oy"], state["ngy"])

    def match_rows(marks, cur, prev, oy, ngy, r0, rows, lay=None):
        # (the interior rows are matched while the halo is in flight: `prev` is read at call time, so a
        # seam row matched too early would see the stale halo and the result would differ from the serial run)
        step = p.patch_sz // 2
        ngx = marks.numel() // ngy
        m = np.zeros(rows * ngx, np.uint64)
        state.update(cur=cur.numpy(), prev=prev.numpy(), oy=oy, ngy=ngy)
        r = O.strip_match(m, cur.numpy().copy(), prev.numpy().copy(), None, SIGMA, p, oy + r0 * step, rows)
        marks[r0 * ngx:(r0 + rows) * ngx].copy_(torch.from_numpy(m.view(np.int64)))
        return r
    return (match, commit, group, match_rows) if state.get("rows") else (match, commit, group)


def _worker(rank, world, port, q, exact=False, rows=False):
    _setup_paths()
    import oracle as O
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    o1, prev, p = _frames(O, synth)
    acc_fn, norm_fn = _callbacks(O, p)
    sf = strips.StripFrame(rank, world, W, H, CH, p.patch_sz, max(p.search_sz_x, p.search_sz_t),
                           torch.device("cpu"), acc_fn, norm_fn,
                           phases=_phases(O, p, {"rows": rows}) if exact else None)
    sf.load(torch.from_numpy(o1), torch.from_numpy(prev))
    sf.step()
    sf.step()  # a second step must give the same result (buffers fully re-initialised)
    y0, y1, rows = sf.own_rows()
    full = torch.zeros((H, W, CH))
    full[y0:y1] = rows
    dist.all_reduce(full)
    if rank == 0:
        q.put(full.numpy())
    dist.destroy_process_group()


def _sequential_strips(O, strips, o1, prev, p, world):
    """The same per-strip algorithm without any communication."""
    halo = max(p.search_sz_x, p.search_sz_t)
    plan = strips.strip_plan(H, p.patch_sz, halo, world)
    step = p.patch_sz // 2
    acc = np.zeros((CH + 1, H, W), np.float32)
    for s in plan:
        a = np.zeros((CH + 1, s["Y1"] - s["Y0"], W), np.float32)
        O.frame_accumulate(a, o1[s["Y0"]:s["Y1"]], prev[s["Y0"]:s["Y1"]], None, SIGMA, p,
                           s["gy0"] * step - s["Y0"], s["gy1"] - s["gy0"])
        acc[:, s["Y0"]:s["Y1"]] += a
    return O.frame_normalize(acc, o1, 0, H)


@pytest.mark.parametrize("world", [2, 3])
def test_strips_over_gloo(world, O, synth):
    import cases
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    o1, prev, p = _frames(O, synth)
    want = _sequential_strips(O, strips, o1, prev, p, world)
    cases.assert_close(got, want, f"{world} ranks vs sequential strips", maxabs=1e-3, rmse=1e-4)
    # against the serial whole-frame order: same class of difference as the
    # reference's own OpenMP row split (mask order at the seams): PSNR-level
    whole = O.filter_frame(o1, prev, None, SIGMA, p)
    clean = O.rgb2opp(synth.clean_frame(W, H, CH, 1))
    assert abs(synth.psnr(got, clean) - synth.psnr(whole, clean)) < 0.05
    assert np.isfinite(got).all()


@pytest.mark.parametrize("world,rows", [(2, False), (3, False), (2, True), (3, True)])
def test_exact_strips_over_gloo_equal_serial_order(world, rows, O, synth):
    """Exact mode: all-gather of the mark words + whole-grid mask replay on every
    rank. The result must equal the serial whole-frame order (not just PSNR-wise).
    rows=True: the interior target rows are matched while the previous-frame halo is still in flight
    (it holds zeros then: a seam row matched too early would change the result)."""
    import cases
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True, rows)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    o1, prev, p = _frames(O, synth)
    whole = O.filter_frame(o1, prev, None, SIGMA, p)
    cases.assert_close(got, whole, f"exact mode, {world} ranks, vs serial whole frame", maxabs=5e-4, rmse=5e-5)


def test_strip_plan_properties():
    strips = importlib.import_module("bwd-nlkalman_amd.strips")
    for (h, psz, halo, world) in [(1080, 8, 10, 8), (1080, 8, 10, 1), (2160, 12, 10, 8), (96, 8, 10, 3)]:
        plan = strips.strip_plan(h, psz, halo, world)
        step = psz // 2
        ngy = (h - psz) // step + 1
        assert plan[0]["gy0"] == 0 and plan[-1]["gy1"] == ngy
        assert plan[0]["own0"] == 0 and plan[-1]["own1"] == h
        for a, b in zip(plan[:-1], plan[1:]):
            assert a["gy1"] == b["gy0"] and a["own1"] == b["own0"]
        for s in plan:
            assert s["Y0"] <= s["own0"] < s["own1"] <= s["Y1"]
            # every candidate / group member of the strip's targets lies inside [Y0, Y1)
            assert s["Y0"] <= max(0, s["gy0"] * step - halo)
            assert s["Y1"] >= min(h, (s["gy1"] - 1) * step + halo + psz)
    with pytest.raises(ValueError):
        strips.strip_plan(64, 8, 10, 8)
