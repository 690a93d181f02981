"""debug: reproduce random case #IT of tests/test_gpu_parity.py::test_randomised_parameters_and_shapes"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as O
built = importlib.import_module("bwd-nlkalman_amd")
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(2024)
for it in range(IT + 1):
    psz = int(rng.choice([4, 6, 8, 8, 8, 10, 12, 12, 16]))
    step = psz // 2
    ch = int(rng.choice([1, 3]))
    w = int(rng.integers(psz, 90)); h = int(rng.integers(psz, 70))
    smoother = rng.random() < 0.25
    wsz_t = int(rng.integers(1, min(15, 3 * step + step - 1) + 1))
    wsz_x = int(rng.integers(1, min(15, 3 * step + step - 1) + 1))
    npx, npt = int(rng.integers(2, 70)), int(rng.integers(2, 70))
    ntagg = int(rng.integers(1, 45))
    over = dict(patch_sz=psz, search_sz_x=wsz_x, search_sz_t=wsz_t, npatches_x=npx, npatches_t=npt, npatches_tagg=ntagg)
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
print(w, h, ch, psz, over, mode, kind, sigma)
po = O.Params(*[getattr(p, k) for k, _ in p._fields_])
fn = O.smooth_frame if smoother else O.filter_frame
r, tr = fn(cur, prev, basic, sigma, po, trace=True)
fg = built.smooth_frame if smoother else built.filter_frame
g = fg(cur, prev, basic, sigma, p)
ng = np.isnan(g)
print("nan ours", ng.sum(), "ref", np.isnan(r).sum(), "prev nan", np.isnan(prev).sum() if prev is not None else 0)
ys, xs = np.nonzero(ng.any(axis=2))
print("nan bbox", ys.min(), ys.max(), xs.min(), xs.max())
act = tr["active"].astype(bool)
ngx = (w - psz) // step + 1
np0 = tr["np0"].reshape(-1, ngx); nsel = tr["nsel"].reshape(-1, ngx); nagg = tr["nagg"].reshape(-1, ngx)
print("np0\n", np0[:6, :8]); print("nsel\n", nsel[:6, :8]); print("nagg\n", nagg[:6, :8]); print("active\n", act.reshape(-1, ngx)[:6, :8].astype(int))
py, px = np.nonzero(np.isnan(prev[..., 0])) if prev is not None else ([], [])
print("prev nan rows", np.unique(py)[:20], "cols", np.unique(px)[:20])
os.environ["NLK_GROUP12_ROWS"] = "1"
g2 = fg(cur, prev, basic, sigma, p)
print("old kernel nan", np.isnan(g2).sum(), "max diff new-vs-old where finite", np.nanmax(np.abs(g - g2)))
