"""Per-step GPU time of the headline loop from in-stream events (no host synchronisation between the steps): where a
short timed loop loses against a long one (development aid; run with gpurun).  python tools/step_trace.py [warmup] [steps] [idle_ms]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
w, h, ch, sigma = 1920, 1080, 3, 20.0
dev = torch.device("cuda", 0)
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 7)
t0_, t1_ = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.rgb2opp(t0_.data_ptr(), w, h, ch); ctx.rgb2opp(t1_.data_ptr(), w, h, ch)
p = pkg.default_params(sigma, pkg.FLT1)
prev, out = torch.empty_like(t0_), torch.empty_like(t1_)
ctx.filter_frame(prev.data_ptr(), t0_.data_ptr(), None, None, w, h, ch, sigma, p)
torch.cuda.synchronize()
if idle:
    time.sleep(idle * 1e-3)
for _ in range(warm):
    ctx.filter_frame(out.data_ptr(), t1_.data_ptr(), prev.data_ptr(), None, w, h, ch, sigma, p)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(steps):
    ctx.filter_frame(out.data_ptr(), t1_.data_ptr(), prev.data_ptr(), None, w, h, ch, sigma, p)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
print(f"warmup {warm}, idle {idle} ms: wall {wall:.4f} ms/step; per step:", " ".join(f"{m:.3f}" for m in ms))
