"""Independent numpy/scipy restatement of nlkalman_filter_frame / _smooth_frame
(SURVEY.md Appendix A / B), written separately from oracle/nlk_oracle.c and
using scipy.fft.dctn(type=2, norm='ortho') for the transform (the published
definition FFTW's REDFT10 + the reference's scaling amount to; reference:
src/nlkalman.c:204-220, 281-298). Pure-Python loops: small frames only.
Used by tests/test_oracle.py to cross-check the C oracle."""
import numpy as np
from scipy.fft import dctn, idctn


def window(psz):
    n = np.arange(psz, dtype=np.float32)
    n2 = np.float32((psz - 1.0) / 2.0)
    x = ((n - n2) / n2 / np.float32(0.4)).astype(np.float32)
    w1 = np.exp(-0.5 * x.astype(np.float64) ** 2).astype(np.float32)
    return np.outer(w1, w1).astype(np.float32)


def _patch(im, x, y, psz):
    return np.transpose(im[y:y + psz, x:x + psz, :], (2, 0, 1)).astype(np.float32)  # [c][hy][hx]


def _dct(p):
    return dctn(p.astype(np.float64), type=2, norm="ortho", axes=(1, 2)).astype(np.float32)


def _idct(p):
    return idctn(p.astype(np.float64), type=2, norm="ortho", axes=(1, 2)).astype(np.float32)


def frame(cur, prev, basic, sigma, P, smoother=False):
    f32 = np.float32
    h, w, ch = cur.shape
    psz, step = P["patch_sz"], P["patch_sz"] // 2
    s2 = f32(sigma) * f32(sigma)
    match = basic if basic is not None else cur
    out = np.zeros_like(cur)
    aggr = np.zeros((h, w), f32)
    mask = np.zeros((h, w), np.int32)
    W = window(psz)
    ntagg = P["npatches_tagg"]

    def valid(x, y):
        return prev is not None and not np.isnan(prev[y:y + psz, x:x + psz, 0]).any()

    for py in range(0, h - psz + 1, step):
        for px in range(0, w - psz + 1, step):
            if mask[py, px]:
                continue
            prev_p = valid(px, py)
            k = P["npatches_t"] if prev_p else P["npatches_x"]
            members, np0, np1 = [], 0, 0
            stats = None
            if k > 1:
                wsz = P["search_sz_t"] if (smoother or prev_p) else P["search_sz_x"]
                x0, x1 = max(px - wsz, 0), min(px + wsz, w - psz) + 1
                y0, y1 = max(py - wsz, 0), min(py + wsz, h - psz) + 1
                tgt = match[py:py + psz, px:px + psz, :]
                cand = []
                for qy in range(y0, y1):
                    for qx in range(x0, x1):
                        e = (match[qy:qy + psz, qx:qx + psz, :] - tgt).reshape(-1)
                        ww = f32(0)
                        for v in e:  # sequential float32 accumulation, hy -> hx -> c
                            ww = f32(ww + f32(v * v))
                        d = f32(ww / f32(psz * psz * ch))
                        cand.append((d if d > 0 else f32(0), len(cand), qx, qy))
                cand.sort(key=lambda t: (t[0], t[1]))
                cand = cand[:min(k, len(cand))]
                A, B, PV = [], [], []
                for (_, _, qx, qy) in cand:
                    pv = prev_p and valid(qx, qy)
                    A.append(_dct(_patch(match, qx, qy, psz)))
                    B.append(_dct(_patch(prev, qx, qy, psz)) if pv else None)
                    PV.append(pv)
                np1 = len(cand)
                np0 = sum(PV)
                A64 = np.array(A, np.float64)
                M1 = A64.mean(0)
                V1 = ((A64 - M1) ** 2).mean(0)
                if np0:
                    B64 = np.array([b for b in B if b is not None], np.float64)
                    Ap = np.array([a for a, pv in zip(A, PV) if pv], np.float64)
                    M0V = B64.mean(0)
                    V0 = ((B64 - M0V) ** 2).mean(0)
                    V01 = ((B64 - Ap) ** 2).mean(0)
                    nag = min(np0, ntagg)
                    M0 = B64[:nag].mean(0) if nag else 0 * M0V
                    idx = [i for i, pv in enumerate(PV) if pv][:nag]
                else:
                    nag = 0 if smoother else min(np1, ntagg)
                    idx = list(range(nag))
                    M0 = V0 = V01 = None
                members = [(cand[i][2], cand[i][3], i) for i in idx]
                stats = (M1, V1, M0, V0, V01, A, B)
            nagg = len(members)
            groups = []
            vp = 0.0
            if nagg:
                M1, V1, M0, V0, V01, A, B = stats
                bsub = 0.0 if basic is not None else float(s2)
                if smoother:
                    a = V1 / (V1 + P["beta_t"] * V01)
                    term = (1 - a * a) * V1 + a * a * np.maximum(V0 - P["beta_t"] * V01, 0)
                elif np0 > 0:
                    v = V0 + np.maximum(0, V01 - bsub)
                    a = v / (v + P["beta_t"] * float(s2))
                    term = (1 - a * a) * v + a * a * float(s2)
                else:
                    v = np.maximum(0, V1 - bsub)
                    a = v / (v + P["beta_x"] * float(s2))
                    term = a * v
                vp = nagg * term.sum()
                for (qx, qy, i) in members:
                    y1_ = _dct(_patch(cur, qx, qy, psz)) if basic is not None else A[i]
                    if smoother:
                        g = (1 - a) * y1_ + a * B[i]
                    elif np0 > 0:
                        g = a * y1_ + (1 - a) * M0
                    else:
                        g = a * y1_ + (1 - a) * M1
                    groups.append((qx, qy, _idct(g)))
            if smoother and np0 == 0:
                groups = [(px, py, _patch(cur, px, py, psz))]
                vp = 0.0
            wgt = f32(1.0) / f32(max(vp, 1e-6))
            if smoother:
                mark = 1 if np0 else 0
            else:
                mark = 0 if (prev is not None and np0 == 0) else 1
            for (qx, qy, g) in groups:
                ww = (wgt * W).astype(f32)
                aggr[qy:qy + psz, qx:qx + psz] += ww
                out[qy:qy + psz, qx:qx + psz, :] += np.transpose(ww[None] * g, (1, 2, 0)).astype(f32)
                mask[qy, qx] += mark
    ok = aggr > 1e-6
    res = np.where(ok[:, :, None], out / np.where(ok, aggr, 1)[:, :, None], cur)
    return res.astype(f32)
