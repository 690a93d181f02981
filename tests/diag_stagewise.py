"""Stage-by-stage GPU vs oracle diagnostic (development aid; run with gpurun).

  python tests/diag_stagewise.py [w h ch sigma]     (lives under tests/: it uses the oracle)
Compares top-k records, active flags and final output for FLT1 spatial,
FLT1 temporal (with NaN holes), FLT2 and SMO1.
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
import oracle as O  # noqa: E402


def to_o(p):
    return O.Params(*[getattr(p, k) for k, _ in p._fields_])


def run_dev(ctx, fn, cur, prev, basic, sigma, p):
    h, w, ch = cur.shape
    d_cur = ctx.upload(cur)
    d_prev = ctx.upload(prev) if prev is not None else None
    d_basic = ctx.upload(basic) if basic is not None else None
    d_out = ctx.alloc(cur.nbytes)
    ctx.set_profiling(True)
    t0 = time.time()
    fn(d_out, d_cur, d_prev, d_basic, w, h, ch, sigma, p)
    ctx.sync()
    dt = time.time() - t0
    out = ctx.download(d_out, cur.shape)
    rec = ctx.read_records()
    tm = ctx.timings()
    for d in (d_cur, d_prev, d_basic, d_out):
        if d:
            ctx.free(d)
    return out, rec, tm, dt


def compare(name, out, rec, ref, tr, tm, dt):
    act_o = tr["active"].astype(bool)
    act_g = rec["active"].astype(bool)
    print(f"== {name}: wall {dt*1e3:.2f} ms, kernels {tm}")
    print(f"   active: oracle {act_o.sum()} gpu {act_g.sum()} mismatches {(act_o != act_g).sum()}")
    both = act_o & act_g
    k = min(tr["topk"].shape[1], rec["topk"].shape[1])
    to, tg = tr["topk"][both][:, :k], rec["topk"][both][:, :k].astype(np.int64)
    nsel = tr["nsel"][both]
    bad = 0
    for i in range(len(nsel)):
        n = nsel[i]
        if not np.array_equal(to[i, :n], tg[i, :n]):
            bad += 1
    print(f"   top-k rows differing: {bad} of {both.sum()}")
    for f in ("nsel", "np0", "nagg"):
        print(f"   {f} mismatches: {(tr[f][both] != rec[f][both]).sum()}", end=";")
    print()
    d = np.abs(out - ref)
    print(f"   out: max-abs {d.max():.3e} rmse {np.sqrt((d**2).mean()):.3e} "
          f"frac>1e-2 {(d > 1e-2).mean():.2e} nan {np.isnan(out).sum()}")


def main():
    a = sys.argv[1:]
    w, h, ch, sigma = (int(a[0]), int(a[1]), int(a[2]), float(a[3])) if len(a) >= 4 else (96, 64, 3, 20.0)
    psz = int(a[4]) if len(a) > 4 else -1
    n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 3)
    o0, o1 = O.rgb2opp(n0), O.rgb2opp(n1)
    ctx = pkg.Context(0)
    p1 = pkg.default_params(sigma, pkg.FLT1, patch_sz=psz)
    p2 = pkg.default_params(sigma, pkg.FLT2, patch_sz=psz)
    ps = pkg.default_params(sigma, pkg.SMO1, patch_sz=psz)
    print("params", p1.as_dict())

    r0, t0 = O.filter_frame(o0, None, None, sigma, to_o(p1), trace=True)
    g0, rec, tm, dt = run_dev(ctx, ctx.filter_frame, o0, None, None, sigma, p1)
    compare("FLT1 spatial", g0, rec, r0, t0, tm, dt)

    flow = np.zeros((h, w, 2), np.float32)
    flow[..., 0] = 2.0 + 0.3 * np.sin(np.arange(w) / 9.0)[None, :]
    flow[..., 1] = 0.25
    occ = np.zeros((h, w), np.float32)
    occ[h // 3:h // 3 + 10, w // 3:w // 3 + 15] = 255
    occ[5, 7] = 255
    prev = O.warp_bicubic(r0, flow, occ)
    # device warp check
    d_im, d_fl, d_oc = ctx.upload(r0), ctx.upload(flow), ctx.upload(occ)
    d_w = ctx.alloc(r0.nbytes)
    ctx.warp_bicubic(d_w, d_im, d_fl, d_oc, w, h, ch)
    gw = ctx.download(d_w, r0.shape)
    same_nan = np.array_equal(np.isnan(gw), np.isnan(prev))
    print(f"== warp: nan pattern equal {same_nan}, max-abs {np.nanmax(np.abs(gw - prev)):.3e}")

    r1, t1 = O.filter_frame(o1, prev, None, sigma, to_o(p1), trace=True)
    g1, rec, tm, dt = run_dev(ctx, ctx.filter_frame, o1, prev, None, sigma, p1)
    compare("FLT1 temporal+NaN", g1, rec, r1, t1, tm, dt)

    r2, t2 = O.filter_frame(o1, prev, r1, sigma, to_o(p2), trace=True)
    g2, rec, tm, dt = run_dev(ctx, ctx.filter_frame, o1, prev, r1, sigma, p2)
    compare("FLT2 temporal+NaN", g2, rec, r2, t2, tm, dt)

    r3, t3 = O.smooth_frame(r0, O.warp_bicubic(r2, flow, occ), None, sigma, to_o(ps), trace=True)
    g3, rec, tm, dt = run_dev(ctx, ctx.smooth_frame, r0, O.warp_bicubic(r2, flow, occ), None, sigma, ps)
    compare("SMO1", g3, rec, r3, t3, tm, dt)

    # colour transforms through the device ABI
    d = ctx.upload(n0)
    ctx.rgb2opp(d, w, h, ch)
    go = ctx.download(d, n0.shape)
    print(f"== rgb2opp max-abs {np.abs(go - o0).max():.3e}")
    ctx.opp2rgb(d, w, h, ch)
    gr = ctx.download(d, n0.shape)
    print(f"== opp2rgb round trip max-abs {np.abs(gr - n0).max():.3e}")


if __name__ == "__main__":
    main()
