#!/usr/bin/env python3
"""bench.py — Mpix/s per frame of the NL-Kalman hot path (nlkalman-flt, FLT1
temporal) on synthetic AWGN frames, BASELINE.json configs[1]: 1920x1080 RGB,
sigma = 20, 8x8 patches.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one nlkalman_filter_frame() of one frame with every input already in
HBM. With N > 1 the frame is split into N row strips of the patch grid (one
process per GPU); per step each rank (1) receives the search halo of the
previous denoised frame from its neighbours (RCCL send/recv over xGMI),
(2) runs the kernels on its strip, (3) sends the accumulator rows it wrote
outside its own rows to their owner and adds what it receives, (4) normalises
its rows. Total work is fixed, so scaling is "strong".
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, ch, sigma, patch, seed)
    "C1": (256, 256, 1, 20.0, 8, 0),
    "C2": (1920, 1080, 3, 20.0, 8, 1),
    "C3": (3840, 2160, 3, 40.0, 12, 2),
}


def strip_plan(h, psz, wsz, world):
    """Row strips of the patch grid: (gy0, gy1, Y0, Y1, own0, own1) per rank."""
    step = psz // 2
    ngy = (h - psz) // step + 1
    plan = []
    for r in range(world):
        gy0 = (ngy * r) // world
        gy1 = (ngy * (r + 1)) // world
        Y0 = max(0, gy0 * step - wsz)
        Y1 = min(h, (gy1 - 1) * step + wsz + psz)
        own0 = gy0 * step if r > 0 else 0
        own1 = gy1 * step if r < world - 1 else h
        plan.append((gy0, gy1, Y0, Y1, own0, own1))
    return plan


def cpu_baseline(O, o1, prev, sigma, p, clean_opp):
    """Oracle ("port") with OpenMP on the host cores: one full frame of the same
    workload (about 25 s of single-core work at 1080p)."""
    nthr = min(O.max_threads(), os.cpu_count() or 1, 100)
    po = O.Params(*[getattr(p, k) for k, _ in p._fields_])
    t0 = time.time()
    out = O.filter_frame(o1, prev, None, sigma, po, nthreads=nthr)
    dt = time.time() - t0
    h, w = o1.shape[:2]
    return out, {"value": round(w * h / dt / 1e6, 4), "unit": "Mpix/s", "cores": nthr,
                 "kind": "port",
                 "sample": f"1 full frame {w}x{h}x{o1.shape[2]} FLT1-temporal, OpenMP over "
                           f"{nthr} threads, {dt:.2f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("bwd-nlkalman_amd")
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    w, h, ch, sigma, psz, seed = WORKLOADS[args.workload]
    p = pkg.default_params(sigma, pkg.FLT1, patch_sz=psz)
    step = p.patch_sz // 2
    ngx = (w - psz) // step + 1
    ngy = (h - psz) // step + 1

    ctx = pkg.Context(local)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    # ---- synthetic inputs (every rank builds the same frames; cheap)
    n0, n1, c1 = synth.noisy_pair(w, h, ch, sigma, seed)
    t_n0 = torch.from_numpy(n0).to(dev)
    t_n1 = torch.from_numpy(n1).to(dev)
    ctx.rgb2opp(t_n0.data_ptr(), w, h, ch)
    ctx.rgb2opp(t_n1.data_ptr(), w, h, ch)
    # previous denoised frame = spatial FLT1 of frame 0, unwarped
    t_prev = torch.empty_like(t_n0)
    ctx.filter_frame(t_prev.data_ptr(), t_n0.data_ptr(), None, None, w, h, ch, sigma, p)
    torch.cuda.synchronize()
    t_out = torch.empty_like(t_n1)

    wsz = max(p.search_sz_x, p.search_sz_t)
    if world == 1:
        def one_step():
            ctx.filter_frame(t_out.data_ptr(), t_n1.data_ptr(), t_prev.data_ptr(), None,
                             w, h, ch, sigma, p)
    else:
        plan = strip_plan(h, psz, wsz, world)
        gy0, gy1, Y0, Y1, own0, own1 = plan[rank]
        hl = Y1 - Y0
        for r in range(world - 1):  # halos must stay inside the neighbour's own rows
            assert plan[r][3] <= plan[r + 1][5] and plan[r + 1][2] >= plan[r][4], "strips too thin"
        s_cur = t_n1[Y0:Y1].contiguous()          # noisy strip + halo (scattered by the host)
        s_prev = torch.empty_like(s_cur)
        s_prev[own0 - Y0:own1 - Y0] = t_prev[own0:own1]   # resident: own rows only
        s_out = torch.empty_like(s_cur)
        acc = torch.empty((ch + 1, hl, w), dtype=torch.float32, device=dev)
        up, dn = rank - 1, rank + 1

        def exchange(ops):
            if ops:
                for r_ in dist.batch_isend_irecv(ops):
                    r_.wait()

        def one_step():
            # (1) input halo of the previous denoised frame
            ops, rbuf_t, rbuf_b = [], None, None
            if up >= 0:
                rbuf_t = torch.empty((own0 - Y0, w, ch), dtype=torch.float32, device=dev)
                n_up = plan[up][3] - own0   # rows of mine the upper rank needs
                ops += [dist.P2POp(dist.isend, s_prev[own0 - Y0:own0 - Y0 + n_up].contiguous(), up),
                        dist.P2POp(dist.irecv, rbuf_t, up)]
            if dn < world:
                rbuf_b = torch.empty((Y1 - own1, w, ch), dtype=torch.float32, device=dev)
                n_dn = own1 - plan[dn][2]   # rows of mine the lower rank needs
                ops += [dist.P2POp(dist.isend, s_prev[own1 - Y0 - n_dn:own1 - Y0].contiguous(), dn),
                        dist.P2POp(dist.irecv, rbuf_b, dn)]
            exchange(ops)
            if rbuf_t is not None:
                s_prev[:own0 - Y0] = rbuf_t
            if rbuf_b is not None:
                s_prev[own1 - Y0:] = rbuf_b
            # (2) kernels on the strip
            acc.zero_()
            ctx.frame_accumulate(acc.data_ptr(), s_cur.data_ptr(), s_prev.data_ptr(), None, w, hl,
                                 ch, sigma, p, gy0 * step - Y0, gy1 - gy0)
            # (3) accumulator rows written outside the own rows go to their owner
            ops, rt, rb = [], None, None
            if up >= 0:
                rt = torch.empty((ch + 1, plan[up][3] - own0, w), dtype=torch.float32, device=dev)
                ops += [dist.P2POp(dist.isend, acc[:, :own0 - Y0].contiguous(), up),
                        dist.P2POp(dist.irecv, rt, up)]
            if dn < world:
                rb = torch.empty((ch + 1, own1 - plan[dn][2], w), dtype=torch.float32, device=dev)
                ops += [dist.P2POp(dist.isend, acc[:, own1 - Y0:].contiguous(), dn),
                        dist.P2POp(dist.irecv, rb, dn)]
            exchange(ops)
            if rt is not None:
                acc[:, own0 - Y0:own0 - Y0 + rt.shape[1]] += rt
            if rb is not None:
                acc[:, own1 - Y0 - rb.shape[1]:own1 - Y0] += rb
            # (4) normalise own rows
            ctx.frame_normalize(s_out.data_ptr(), acc.data_ptr(), s_cur.data_ptr(), w, hl, ch,
                                own0 - Y0, own1 - Y0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    ctx.set_profiling(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # gather the strips on rank 0 for the quality check
        full = torch.zeros_like(t_n1)
        full[own0:own1] = s_out[own0 - Y0:own1 - Y0]
        dist.all_reduce(full)
        t_out = full

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = w * h / (dt / args.steps) / 1e6
        out = t_out.cpu().numpy()
        # dominant kernel and its roofline (algorithmic bytes: DESIGN.md §4)
        k = p.npatches_t
        ngrid = ngx * ngy
        alg = {"match": w * h * ch * 4 + ngrid * k * 4,
               "group": 2 * w * h * ch * 4 + (ch + 1) * w * h * 4 + ngrid * k * 4}
        dom = "group" if tm["group_ms"] >= tm["match_ms"] else "match"
        dur = tm[dom + "_ms"] * 1e-3
        gbs = alg[dom] / world / dur / 1e9 if dur > 0 else 0.0
        roof = {"kernel": "k_group" if dom == "group" else "k_bm_topk", "bound": "hbm",
                "achieved": round(gbs, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(gbs / 8000.0, 6), "traffic": None,
                "launch_ms": round(tm[dom + "_ms"], 4),
                "algorithmic_bytes_per_launch": alg[dom] // world,
                "note": "path is VALU/LDS-bound (about 1e3 flop/B), see DESIGN.md"}
        res = {"metric": "Mpix/s per frame (nlkalman-flt, 1080p sigma=20)"
               if args.workload == "C2" else f"Mpix/s per frame (nlkalman-flt, {args.workload})",
               "value": round(value, 3), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{args.workload}: {w}x{h}x{ch} synthetic AWGN sigma={sigma:g}, "
                                      f"FLT1 temporal (deno0 = spatial FLT1 of frame 0, bsic1=NULL), "
                                      f"patch {psz}, defaults of nlkalman_default_params",
                          "parallelism": f"row strips x{world}" if world > 1 else "single GPU",
                          "mask_order": "serial-exact" if world == 1 else "per-strip (as the reference's OpenMP row split)"},
               "kernels_ms": {k_: round(v, 4) for k_, v in tm.items()},
               "roofline": roof}
        if not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            o1 = t_n1.cpu().numpy()
            prev = t_prev.cpu().numpy()
            ref, cb = cpu_baseline(O, o1, prev, sigma, p, None)
            res["cpu_baseline"] = cb
            c1o = O.rgb2opp(c1)
            res["psnr_gpu_db"] = round(synth.psnr(O.opp2rgb(out), c1), 4)
            res["psnr_cpu_db"] = round(synth.psnr(O.opp2rgb(ref), c1), 4)
            res["psnr_delta_db"] = round(res["psnr_gpu_db"] - res["psnr_cpu_db"], 4)
            res["speedup_vs_cpu"] = round(value / cb["value"], 1)
            del c1o
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
