"""ctypes bindings of the CPU oracle (oracle/nlk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLT1, FLT2, SMO1 = 0, 1, 2


class Params(C.Structure):
    # reference: src/nlkalman.h:22-37 (K_SIMILAR_PATCHES layout)
    _fields_ = [("patch_sz", C.c_int), ("search_sz_x", C.c_int),
                ("search_sz_t", C.c_int), ("npatches_x", C.c_int),
                ("npatches_t", C.c_int), ("npatches_tagg", C.c_int),
                ("dista_lambda", C.c_float), ("beta_x", C.c_float),
                ("beta_t", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Trace(C.Structure):
    _fields_ = [("kmax", C.c_int), ("gmax", C.c_int),
                ("topk", C.POINTER(C.c_int)), ("gcoords", C.POINTER(C.c_int)),
                ("nsel", C.POINTER(C.c_int)), ("np0", C.POINTER(C.c_int)),
                ("np1", C.POINTER(C.c_int)), ("nagg", C.POINTER(C.c_int)),
                ("active", C.POINTER(C.c_int)), ("vp", C.POINTER(C.c_float)),
                ("aggr", C.POINTER(C.c_float))]


def build():
    """Compile oracle/libnlk_oracle.so if missing or stale."""
    so = os.path.join(_HERE, "libnlk_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("nlk_oracle.c", "tvl1_oracle.c", "ms_oracle.c", "nlk_oracle.h")]
    if (not os.path.exists(so)
            or os.path.getmtime(so) < max(os.path.getmtime(s) for s in src)):
        subprocess.check_call(["make", "-C", _HERE, "libnlk_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        L.nlko_default_params.argtypes = [C.POINTER(Params), C.c_float, C.c_int]
        L.nlko_window.argtypes = [fp, C.c_int]
        for f in (L.nlko_rgb2opp, L.nlko_opp2rgb):
            f.argtypes = [fp, C.c_int, C.c_int, C.c_int]
        L.nlko_warp_bicubic.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int]
        L.nlko_awgn.argtypes = [fp, C.c_long, C.c_float, C.c_uint32]
        L.nlko_dct_basis.argtypes = [fp, C.c_int]
        L.nlko_dct2.argtypes = [fp, C.c_int, C.c_int, C.c_int]
        for f in (L.nlko_filter_frame, L.nlko_smooth_frame):
            f.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float,
                          C.POINTER(Params), C.c_int, C.POINTER(Trace)]
        L.nlko_frame_accumulate.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.POINTER(Params), C.c_int, C.c_int, C.c_int]
        L.nlko_frame_normalize.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
        L.nlko_strip_match.argtypes = [u64p, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.nlko_mask_commit.argtypes = [u64p, C.c_int, C.c_int, C.c_int, u8p]
        L.nlko_strip_group.argtypes = [fp, u8p, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.POINTER(Params), C.c_int, C.c_int, C.c_int]
        L.nlko_max_threads.restype = C.c_int
        i, f = C.c_int, C.c_float
        L.tvl1o_bicubic_at.argtypes = [fp, f, f, i, i, i]
        L.tvl1o_bicubic_at.restype = f
        L.tvl1o_forward_gradient.argtypes = [fp, fp, fp, i, i]
        L.tvl1o_centered_gradient.argtypes = [fp, fp, fp, i, i]
        L.tvl1o_divergence.argtypes = [fp, fp, fp, i, i]
        L.tvl1o_gaussian.argtypes = [fp, i, i, C.c_double]
        L.tvl1o_zoom_size.argtypes = [i, i, C.POINTER(i), C.POINTER(i), f]
        L.tvl1o_zoom_out.argtypes = [fp, fp, i, i, f]
        L.tvl1o_zoom_in.argtypes = [fp, fp, i, i, i, i]
        L.tvl1o_flow_scale.argtypes = [fp, fp, fp, fp, i, i, f, f, f, i, f, C.POINTER(i)]
        L.tvl1o_flow_scale.restype = i
        L.tvl1o_normalize.argtypes = [fp, fp, fp, fp, i]
        L.tvl1o_auto_scales.argtypes = [i, i, i, f]
        L.tvl1o_auto_scales.restype = i
        L.tvl1o_flow.argtypes = [fp, fp, fp, fp, i, i, f, f, f, i, i, f, i, f]
        L.tvl1o_occlusion_mask.argtypes = [fp, fp, i, i, f]
        L.mso_image_dct.argtypes = [fp, i, i, i, i]
        _LIB = L
    return _LIB


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _img(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 2:
        a = a[:, :, None]
    return a


def default_params(sigma, mode, **over):
    p = Params(-1, -1, -1, -1, -1, -1, -1.0, -1.0, -1.0)
    for k, v in over.items():
        setattr(p, k, v)
    lib().nlko_default_params(C.byref(p), float(sigma), int(mode))
    return p


def window(psz):
    W = np.zeros((psz, psz), np.float32)
    lib().nlko_window(_fp(W), psz)
    return W


def rgb2opp(im):
    a = _img(im).copy()
    lib().nlko_rgb2opp(_fp(a), a.shape[1], a.shape[0], a.shape[2])
    return a


def opp2rgb(im):
    a = _img(im).copy()
    lib().nlko_opp2rgb(_fp(a), a.shape[1], a.shape[0], a.shape[2])
    return a


def warp_bicubic(im, flow, occ=None):
    im = _img(im)
    h, w, ch = im.shape
    flow = np.ascontiguousarray(flow, np.float32)
    occ = None if occ is None else np.ascontiguousarray(occ, np.float32)
    out = np.empty_like(im)
    lib().nlko_warp_bicubic(_fp(out), _fp(im), _fp(flow), _fp(occ), w, h, ch)
    return out


def awgn(im, sigma, seed):
    a = np.ascontiguousarray(im, np.float32).copy()
    lib().nlko_awgn(_fp(a), a.size, float(sigma), int(seed))
    return a


def dct_basis(n):
    Cm = np.zeros((n, n), np.float32)
    lib().nlko_dct_basis(_fp(Cm), n)
    return Cm


def dct2(planes, inverse=False):
    a = np.ascontiguousarray(planes, np.float32).copy()
    n = a.shape[-1]
    assert a.shape[-2] == n
    lib().nlko_dct2(_fp(a), n, a.size // (n * n), int(inverse))
    return a


def grid_shape(w, h, psz):
    step = psz // 2
    return ((w - psz) // step + 1 if w >= psz else 0), ((h - psz) // step + 1 if h >= psz else 0)


def _run(fn, cur, prev, basic, sigma, params, nthreads, trace):
    cur, prev, basic = _img(cur), _img(prev), _img(basic)
    h, w, ch = cur.shape
    out = np.empty_like(cur)
    tr = None
    res = {}
    if trace:
        ngx, ngy = grid_shape(w, h, params.patch_sz)
        ng = ngx * ngy
        kmax = max(params.npatches_x, params.npatches_t, 1)
        gmax = max(params.npatches_tagg, 1)
        res = dict(
            topk=np.full((ng, kmax), -1, np.int32),
            gcoords=np.full((ng, gmax), -1, np.int32),
            nsel=np.zeros(ng, np.int32), np0=np.zeros(ng, np.int32),
            np1=np.zeros(ng, np.int32), nagg=np.zeros(ng, np.int32),
            active=np.zeros(ng, np.int32), vp=np.zeros(ng, np.float32),
            aggr=np.zeros((h, w), np.float32))
        tr = Trace(kmax, gmax, _ip(res["topk"]), _ip(res["gcoords"]),
                   _ip(res["nsel"]), _ip(res["np0"]), _ip(res["np1"]),
                   _ip(res["nagg"]), _ip(res["active"]), _fp(res["vp"]),
                   _fp(res["aggr"]))
        res["grid"] = (ngx, ngy)
    fn(_fp(out), _fp(cur), _fp(prev), _fp(basic), w, h, ch, float(sigma),
       C.byref(params), int(nthreads), None if tr is None else C.byref(tr))
    return (out, res) if trace else out


def filter_frame(nisy1, deno0, bsic1, sigma, params, nthreads=1, trace=False):
    return _run(lib().nlko_filter_frame, nisy1, deno0, bsic1, sigma, params,
                nthreads, trace)


def smooth_frame(filt1, smoo0, bsic1, sigma, params, nthreads=1, trace=False):
    return _run(lib().nlko_smooth_frame, filt1, smoo0, bsic1, sigma, params,
                nthreads, trace)


def frame_accumulate(acc, cur, prev, basic, sigma, params, oy, ngy, smoother=False):
    """Row-strip form: adds into the planar accumulator acc[(ch+1), h, w] in place."""
    cur, prev, basic = _img(cur), _img(prev), _img(basic)
    h, w, ch = cur.shape
    assert acc.dtype == np.float32 and acc.shape == (ch + 1, h, w) and acc.flags.c_contiguous
    lib().nlko_frame_accumulate(_fp(acc), _fp(cur), _fp(prev), _fp(basic), w, h, ch, float(sigma),
                                C.byref(params), int(oy), int(ngy), int(smoother))


def frame_normalize(acc, cur, y0, y1):
    cur = _img(cur)
    h, w, ch = cur.shape
    out = np.zeros_like(cur)
    lib().nlko_frame_normalize(_fp(out), _fp(acc), _fp(cur), w, h, ch, int(y0), int(y1))
    return out


def strip_match(marks, cur, prev, basic, sigma, params, oy, ngy, smoother=False):
    """Three-phase strip form, phase 1: fills marks[ngy*ngx] (uint64), returns the reach R."""
    cur, prev, basic = _img(cur), _img(prev), _img(basic)
    h, w, ch = cur.shape
    r = C.c_int()
    lib().nlko_strip_match(marks.ctypes.data_as(C.POINTER(C.c_uint64)), _fp(cur), _fp(prev), _fp(basic),
                           w, h, ch, float(sigma), C.byref(params), int(oy), int(ngy), int(smoother),
                           C.byref(r))
    return r.value


def mask_commit(marks, ngx, ngy, reach):
    active = np.zeros(ngx * ngy, np.uint8)
    lib().nlko_mask_commit(marks.ctypes.data_as(C.POINTER(C.c_uint64)), ngx, ngy, reach,
                           active.ctypes.data_as(C.POINTER(C.c_uint8)))
    return active


def strip_group(acc, active, cur, prev, basic, sigma, params, oy, ngy, smoother=False):
    cur, prev, basic = _img(cur), _img(prev), _img(basic)
    h, w, ch = cur.shape
    active = np.ascontiguousarray(active, np.uint8)
    lib().nlko_strip_group(_fp(acc), active.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(cur), _fp(prev),
                           _fp(basic), w, h, ch, float(sigma), C.byref(params), int(oy), int(ngy),
                           int(smoother))


def max_threads():
    return lib().nlko_max_threads()


RELEASE_FLAGS = "-O3 -ffast-math -fno-finite-math-only -fopenmp"   # = oracle/Makefile RELEASE_FLAGS
STRICT_FLAGS = "-O2 -ffp-contract=off -fno-fast-math -fopenmp"      # = oracle/Makefile CFLAGS


def release_filter_frame():
    """nlko_filter_frame of the build with the reference's release flags (oracle/Makefile:
    libnlk_oracle_release.so) - timed by bench.py's cpu_baseline leg, never used as a checker."""
    so = os.path.join(_HERE, "libnlk_oracle_release.so")
    src = [os.path.join(_HERE, f) for f in ("nlk_oracle.c", "nlk_oracle.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libnlk_oracle_release.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    L.nlko_filter_frame.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.POINTER(Params), C.c_int, C.POINTER(Trace)]
    return L.nlko_filter_frame


def filter_frame_with(fn, nisy1, deno0, bsic1, sigma, params, nthreads=1):
    """filter_frame through another build's entry point (release_filter_frame())."""
    return _run(fn, nisy1, deno0, bsic1, sigma, params, nthreads, False)


# ---- dual TV-L1 optical flow (tvl1_oracle.c; reference: lib/tvl1flow/)
TVL1_DEFAULTS = dict(tau=0.25, lam=0.15, theta=0.3, nscales=100, fscale=0, zfactor=0.5,
                     nwarps=5, epsilon=0.01)  # reference: lib/tvl1flow/main.c:26-35


def tvl1_auto_scales(w, h, nscales=100, zfactor=0.5):
    return lib().tvl1o_auto_scales(w, h, nscales, zfactor)


def tvl1_flow(i0, i1, **kw):
    """Flow from gray image i0 to i1 (h, w float32) -> (u, v), parameters as the
    reference's command line (the number of scales is capped by the image size)."""
    p = dict(TVL1_DEFAULTS, **kw)
    i0 = np.ascontiguousarray(i0, np.float32)
    i1 = np.ascontiguousarray(i1, np.float32)
    h, w = i0.shape
    ns = tvl1_auto_scales(w, h, p["nscales"], p["zfactor"])
    fs = min(p["fscale"], ns)
    u, v = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    lib().tvl1o_flow(_fp(i0), _fp(i1), _fp(u), _fp(v), w, h, p["tau"], p["lam"], p["theta"], ns, fs,
                     p["zfactor"], p["nwarps"], p["epsilon"])
    return u, v


def tvl1_gray(im):
    """Luminance the reference's reader hands to the flow for a colour file: double sum, one
    rounding (lib/iio/iio.c:1048-1056 via iio_read_image_float, :3993-3994)."""
    im = np.asarray(im, np.float32)
    if im.ndim == 2 or im.shape[2] == 1:
        return np.ascontiguousarray(im.reshape(im.shape[0], im.shape[1]))
    c = im.astype(np.float64)
    return np.ascontiguousarray((.299 * c[..., 0] + .587 * c[..., 1] + .114 * c[..., 2]).astype(np.float32))


def tvl1_occlusion_mask(flow, th):
    """flow (h, w, 2) -> mask (h, w) of 0 / 255 (|divergence| > th)."""
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    m = np.zeros((h, w), np.float32)
    lib().tvl1o_occlusion_mask(_fp(flow), _fp(m), w, h, th)
    return m


# ---- multiscale wrapper (ms_oracle.c; reference: lib/multiscale/)
def ms_dct(im, inverse=False):
    """Whole-image DCT of the multiscale tools (dct_inplace / idct_inplace)."""
    a = np.array(_img(im), np.float32, copy=True)
    h, w, ch = a.shape
    lib().mso_image_dct(_fp(a), w, h, ch, int(inverse))
    return a


def ms_decompose(im, levels, ratio=2.0):
    """decompose (reference: lib/multiscale/decompose.cpp:27-56): level i = inverse DCT of the
    top-left h_i x w_i block of the image's DCT, sizes divided by `ratio` (truncated) per level."""
    co = ms_dct(im)
    h, w = co.shape[:2]
    out = []
    for _ in range(levels):
        out.append(ms_dct(np.ascontiguousarray(co[:h, :w]), inverse=True))
        w = int(w / np.float32(ratio))
        h = int(h / np.float32(ratio))
    return out


def ms_recompose(levels, factor=0.8):
    """recompose (reference: lib/multiscale/recompose.cpp:24-56): the low frequencies of level 0
    are replaced by those of the coarser levels (the first rows*factor x cols*factor of each)."""
    out = ms_dct(levels[0])
    for im in levels[1:]:
        co = ms_dct(im)
        f = np.float32(factor)
        bh = int(np.ceil(np.float32(co.shape[0]) * f))
        bw = int(np.ceil(np.float32(co.shape[1]) * f))
        out[:bh, :bw] = co[:bh, :bw]
    return ms_dct(out, inverse=True)
