"""Aggregate throughput of several independent sequences on ONE GPU (one context = one HIP stream
each, one host thread each). Development aid; run with gpurun:  python tools/seq_streams.py [n ...]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
seq = importlib.import_module("bwd-nlkalman_amd.sequence")
w, h, ch, sigma, frames = 1920, 1080, 3, 20.0, 12
cleans = [synth.clean_frame(w, h, ch, t) for t in range(4)]
noisy = [synth.awgn(cleans[t], sigma, 10 + t) for t in range(4)]


def worker(ctx, dfr, n, out, i):
    sf = seq.SequenceFilter(ctx, w, h, ch, sigma, keep_history=False)
    sf.push(dfr[0])
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(n):
        sf.push(dfr[(k + 1) % 4])
    ctx.sync()
    out[i] = time.perf_counter() - t0


for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    ctxs = [pkg.Context(0) for _ in range(ns)]
    dfrs = [[c.upload(f) for f in noisy] for c in ctxs]
    out = [0.0] * ns
    th = [threading.Thread(target=worker, args=(ctxs[i], dfrs[i], frames, out, i)) for i in range(ns)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    print(f"{ns} sequence(s): {ns * frames / max(out):.1f} frames/s aggregate ({max(out) / frames * 1e3:.2f} ms per frame per sequence, wall {wall:.2f} s)")
