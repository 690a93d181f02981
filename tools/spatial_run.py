import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("bwd-nlkalman_amd")
synth = importlib.import_module("bwd-nlkalman_amd.synth")
w, h, ch, sigma = 1920, 1080, 3, 20.0
n0, n1, _ = synth.noisy_pair(w, h, ch, sigma, 7)
ctx = pkg.Context(0)
d0 = ctx.upload(n0); ctx.rgb2opp(d0, w, h, ch)
o0 = ctx.alloc(n0.nbytes)
p1 = pkg.default_params(sigma, pkg.FLT1)
for _ in range(3):
    ctx.filter_frame(o0, d0, None, None, w, h, ch, sigma, p1)
ctx.sync()
