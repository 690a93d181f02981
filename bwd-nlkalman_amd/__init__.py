"""bwd-nlkalman_amd — MI355X-native NL-Kalman per-frame hot path.

Python is only a thin ctypes mirror of the two C interfaces of the product:

* ``libnlkalman.so``  — the drop-in C API of include/nlkalman.h (same symbols as
  the reference's src/nlkalman.h:14-53), host pointers in / out;
* ``libnlk_hip.so``   — the C-ABI of include/nlk_hip.h over the HIP kernels,
  device pointers in / out (used by bench.py and the multi-GPU driver).

There is no CPU fallback here: every function that touches pixels needs the
HIP library and a GPU, and raises loudly otherwise.  (The directory name holds
a hyphen: import it with ``importlib.import_module("bwd-nlkalman_amd")``.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FLT1, FLT2, SMO1 = 0, 1, 2

HIP_SYMBOLS = [
    "nlk_device_count", "nlk_ctx_create", "nlk_ctx_destroy", "nlk_last_error",
    "nlk_ctx_set_profiling", "nlk_ctx_get_timings", "nlk_ctx_set_stream",
    "nlk_ctx_get_stream", "nlk_ctx_use_own_stream", "nlk_dev_alloc", "nlk_dev_free", "nlk_h2d", "nlk_d2h",
    "nlk_d2d", "nlk_sync", "nlk_host_alloc", "nlk_host_free", "nlk_dev_rgb2opp", "nlk_dev_opp2rgb",
    "nlk_dev_warp_bicubic", "nlk_dev_filter_frame", "nlk_dev_smooth_frame",
    "nlk_dev_frame_accumulate", "nlk_dev_frame_normalize", "nlk_ctx_read_records",
    "nlk_dev_strip_match", "nlk_dev_strip_match_rows", "nlk_dev_mask_commit", "nlk_dev_strip_group",
    "nlk_tvl1_default_params", "nlk_tvl1_scales", "nlk_dev_tvl1_flow", "nlk_dev_gray",
    "nlk_dev_occlusion_mask", "nlk_dev_image_dct", "nlk_dev_copy_block", "nlk_host_tables", "nlk_ctx_set_deterministic", "nlk_dev_zero", "nlk_dev_add", "nlk_dev_copy_peer",
    "nlk_filter_frame_host", "nlk_smooth_frame_host", "nlk_dev_strip_match_part", "nlk_ctx_reload_switches", "nlk_ctx_set_strip_accumulator",
    "nlk_strips_create", "nlk_strips_destroy", "nlk_strips_last_error", "nlk_rccl_unique_id", "nlk_strips_rccl_init",
    "nlk_strips_transport", "nlk_strips_load", "nlk_strips_set_options", "nlk_strips_step", "nlk_strips_sync",
    "nlk_strips_own_rows", "nlk_strips_ctx", "nlk_strips_geometry", "nlk_strips_stats", "nlk_strips_set_dry_run",
    "nlk_dev_strip_commit_group", "nlk_ctx_flush_active",
]
API_SYMBOLS = [
    "rgb2opp", "opp2rgb", "warp_bicubic", "nlkalman_default_params",
    "nlkalman_filter_frame", "nlkalman_smooth_frame",
]
TVL1_SYMBOLS = ["Dual_TVL1_optic_flow_multiscale"]  # include/tvl1flow.h, exported by libnlkalman.so


class Tvl1Params(C.Structure):
    """struct nlk_tvl1_params (include/nlk_hip.h; reference: lib/tvl1flow/main.c:26-35)."""
    _fields_ = [("tau", C.c_float), ("lam", C.c_float), ("theta", C.c_float),
                ("nscales", C.c_int), ("fscale", C.c_int), ("zfactor", C.c_float),
                ("nwarps", C.c_int), ("epsilon", C.c_float)]


class Params(C.Structure):
    """struct nlkalman_params (reference: src/nlkalman.h:22-37)."""
    _fields_ = [("patch_sz", C.c_int), ("search_sz_x", C.c_int),
                ("search_sz_t", C.c_int), ("npatches_x", C.c_int),
                ("npatches_t", C.c_int), ("npatches_tagg", C.c_int),
                ("dista_lambda", C.c_float), ("beta_x", C.c_float),
                ("beta_t", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Timings(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("layout_ms", "match_ms", "commit_ms",
                                          "group_ms", "normalize_ms", "total_ms")]


def build(force=False):
    """Compile the in-tree libraries with hipcc/gcc (gfx950). No GPU needed."""
    args = ["make", "-j8", "-C", _HERE, "all"]
    if force:
        args.insert(1, "-B")
    subprocess.check_call(args)


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{name} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP path is mandatory, there is no CPU fallback)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_hip = None
_api = None


def hip():
    """libnlk_hip.so with argtypes set."""
    global _hip
    if _hip is None:
        L = _load("libnlk_hip.so")
        vp, fp, i, f = C.c_void_p, C.c_void_p, C.c_int, C.c_float
        L.nlk_device_count.restype = i
        L.nlk_ctx_create.argtypes = [C.POINTER(vp), i]
        L.nlk_ctx_destroy.argtypes = [vp]
        L.nlk_ctx_destroy.restype = None
        L.nlk_last_error.argtypes = [vp]
        L.nlk_last_error.restype = C.c_char_p
        L.nlk_ctx_set_profiling.argtypes = [vp, i]
        L.nlk_ctx_get_timings.argtypes = [vp, C.POINTER(Timings)]
        L.nlk_ctx_set_stream.argtypes = [vp, vp]
        L.nlk_ctx_use_own_stream.argtypes = [vp]
        L.nlk_ctx_get_stream.argtypes = [vp]
        L.nlk_ctx_get_stream.restype = vp
        L.nlk_dev_alloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
        L.nlk_dev_free.argtypes = [vp, vp]
        for fn in (L.nlk_h2d, L.nlk_d2h, L.nlk_d2d):
            fn.argtypes = [vp, vp, vp, C.c_size_t]
        L.nlk_sync.argtypes = [vp]
        for fn in (L.nlk_dev_rgb2opp, L.nlk_dev_opp2rgb):
            fn.argtypes = [vp, fp, i, i, i]
        L.nlk_dev_warp_bicubic.argtypes = [vp, fp, fp, fp, fp, i, i, i]
        for fn in (L.nlk_dev_filter_frame, L.nlk_dev_smooth_frame):
            fn.argtypes = [vp, fp, fp, fp, fp, i, i, i, f, C.POINTER(Params)]
        L.nlk_dev_frame_accumulate.argtypes = [vp, fp, fp, fp, fp, i, i, i, f,
                                               C.POINTER(Params), i, i, i]
        L.nlk_dev_frame_normalize.argtypes = [vp, fp, fp, fp, i, i, i, i, i]
        L.nlk_dev_strip_match.argtypes = [vp, fp, fp, fp, i, i, i, f, C.POINTER(Params), i, i, i,
                                          vp, C.POINTER(i)]
        L.nlk_dev_strip_match_rows.argtypes = [vp, fp, fp, fp, i, i, i, f, C.POINTER(Params), i, i, i, i, i,
                                               vp, C.POINTER(i)]
        L.nlk_dev_strip_match_part.argtypes = [vp, fp, fp, fp, i, i, i, f, C.POINTER(Params), i, i, i, i, i, i, i, i, i,
                                               vp, C.POINTER(i)]
        L.nlk_dev_mask_commit.argtypes = [vp, vp, i, i, i, vp]
        L.nlk_dev_strip_group.argtypes = [vp, fp, vp]
        L.nlk_dev_strip_commit_group.argtypes = [vp, fp, vp, i, i, i, i, vp]
        L.nlk_ctx_read_records.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i),
                                           vp, vp, vp, vp, vp, vp]
        L.nlk_tvl1_default_params.argtypes = [C.POINTER(Tvl1Params)]
        L.nlk_tvl1_default_params.restype = None
        L.nlk_tvl1_scales.argtypes = [i, i, i, f]
        L.nlk_dev_tvl1_flow.argtypes = [vp, fp, fp, fp, i, i, C.POINTER(Tvl1Params), C.POINTER(i)]
        L.nlk_dev_gray.argtypes = [vp, fp, fp, i, i, i]
        L.nlk_dev_occlusion_mask.argtypes = [vp, fp, fp, i, i, f]
        L.nlk_dev_image_dct.argtypes = [vp, fp, i, i, i, i]
        L.nlk_dev_copy_block.argtypes = [vp, fp, i, fp, i, i, i, i]
        L.nlk_host_tables.argtypes = [i, vp, vp, vp]
        L.nlk_ctx_set_deterministic.argtypes = [vp, i]
        L.nlk_ctx_reload_switches.argtypes = [vp]
        L.nlk_ctx_set_strip_accumulator.argtypes = [vp, vp]
        L.nlk_strips_create.argtypes = [C.POINTER(vp), i, C.POINTER(i), i, i, i, i, i, C.c_float, C.POINTER(Params), i, i]
        L.nlk_strips_destroy.argtypes = [vp]
        L.nlk_strips_destroy.restype = None
        L.nlk_strips_last_error.argtypes = [vp]
        L.nlk_strips_last_error.restype = C.c_char_p
        L.nlk_rccl_unique_id.argtypes = [vp]
        L.nlk_strips_rccl_init.argtypes = [vp, vp]
        L.nlk_strips_transport.argtypes = [vp]
        L.nlk_strips_transport.restype = C.c_char_p
        L.nlk_strips_load.argtypes = [vp, i, vp, vp]
        L.nlk_strips_set_options.argtypes = [vp, i, i, i]
        L.nlk_strips_step.argtypes = [vp]
        L.nlk_strips_set_dry_run.argtypes = [vp, i]
        L.nlk_strips_sync.argtypes = [vp]
        L.nlk_strips_own_rows.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.nlk_strips_ctx.argtypes = [vp, i]
        L.nlk_strips_ctx.restype = vp
        L.nlk_strips_geometry.argtypes = [vp, i, C.POINTER(i)]
        L.nlk_strips_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i)]
        _hip = L
    return _hip


def api():
    """libnlkalman.so (drop-in C API) with argtypes set."""
    global _api
    if _api is None:
        hip()  # dependency, RTLD_GLOBAL
        L = _load("libnlkalman.so")
        fp, i, f = C.POINTER(C.c_float), C.c_int, C.c_float
        for fn in (L.rgb2opp, L.opp2rgb):
            fn.argtypes = [fp, i, i, i]
            fn.restype = None
        L.warp_bicubic.argtypes = [fp, fp, fp, fp, i, i, i]
        L.warp_bicubic.restype = None
        L.nlkalman_default_params.argtypes = [C.POINTER(Params), f, i]
        L.nlkalman_default_params.restype = None
        for fn in (L.nlkalman_filter_frame, L.nlkalman_smooth_frame):
            fn.argtypes = [fp, fp, fp, fp, i, i, i, f, Params, i]
            fn.restype = None
        _api = L
    return _api


class NlkError(RuntimeError):
    pass


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _img(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a[:, :, None] if a.ndim == 2 else a


# ---------------------------------------------------------------- drop-in API

def tvl1_params(w, h, **over):
    """nlk_tvl1_default_params + the scale count the reference's command line would use
    for a w x h image (fields may be overridden, e.g. lam=0.4, fscale=1)."""
    p = Tvl1Params()
    hip().nlk_tvl1_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    p.nscales = hip().nlk_tvl1_scales(w, h, p.nscales, p.zfactor)
    p.fscale = min(p.fscale, p.nscales)
    return p


def default_params(sigma, mode, **over):
    """nlkalman_default_params (reference: src/nlkalman.c:426-487); host-only."""
    p = Params(-1, -1, -1, -1, -1, -1, -1.0, -1.0, -1.0)
    for k, v in over.items():
        setattr(p, k, v)
    api().nlkalman_default_params(C.byref(p), float(sigma), int(mode))
    return p


def filter_frame(nisy1, deno0, bsic1, sigma, params):
    """nlkalman_filter_frame through the drop-in C API (host arrays)."""
    nisy1, deno0, bsic1 = _img(nisy1), _img(deno0), _img(bsic1)
    h, w, ch = nisy1.shape
    out = np.empty_like(nisy1)
    api().nlkalman_filter_frame(_fp(out), _fp(nisy1), _fp(deno0), _fp(bsic1), w, h, ch,
                                float(sigma), params, 0)
    return out


def smooth_frame(filt1, smoo0, bsic1, sigma, params):
    filt1, smoo0, bsic1 = _img(filt1), _img(smoo0), _img(bsic1)
    h, w, ch = filt1.shape
    out = np.empty_like(filt1)
    api().nlkalman_smooth_frame(_fp(out), _fp(filt1), _fp(smoo0), _fp(bsic1), w, h, ch,
                                float(sigma), params, 0)
    return out


def rgb2opp(im):
    a = _img(im).copy()
    api().rgb2opp(_fp(a), a.shape[1], a.shape[0], a.shape[2])
    return a


def opp2rgb(im):
    a = _img(im).copy()
    api().opp2rgb(_fp(a), a.shape[1], a.shape[0], a.shape[2])
    return a


def warp_bicubic(im, flow, occ=None):
    im = _img(im)
    h, w, ch = im.shape
    flow = np.ascontiguousarray(flow, np.float32)
    occ = None if occ is None else np.ascontiguousarray(occ, np.float32)
    out = np.empty_like(im)
    api().warp_bicubic(_fp(out), _fp(im), _fp(flow), _fp(occ), w, h, ch)
    return out


def host_tables(psz):
    """The DCT basis and aggregation window a frame call uploads for this patch size, and the 12x12
    matrix the 12-point flow graph of k_dct12.h applies (nlk_host_tables; no device needed)."""
    b, w, b12 = (np.zeros((psz, psz), np.float32), np.zeros((psz, psz), np.float32),
                 np.zeros((12, 12), np.float32))
    rc = hip().nlk_host_tables(psz, b.ctypes.data, w.ctypes.data, b12.ctypes.data)
    if rc:
        raise NlkError(hip().nlk_last_error(None).decode())
    return b, w, b12


def reload_switches():
    """Re-read the NLK_* environment switches for every live context of this process (they are read once,
    at context creation: include/nlk_hip.h). Only the tests need this."""
    if _hip is not None:
        _hip.nlk_ctx_reload_switches(None)


# ------------------------------------------------------- device-resident C-ABI

class Context:
    """nlk_ctx wrapper: device pointers are plain ints (e.g. tensor.data_ptr())."""

    def __init__(self, device=0):
        self.L = hip()
        self.h = C.c_void_p()
        rc = self.L.nlk_ctx_create(C.byref(self.h), int(device))
        if rc:
            raise NlkError(self.L.nlk_last_error(None).decode())

    @classmethod
    def from_handle(cls, handle):
        """A view of a context somebody else owns (a strip's: nlk_strips_ctx); never destroyed from here."""
        c = cls.__new__(cls)
        c.L, c.h, c._borrowed = hip(), C.c_void_p(handle), True
        return c

    def close(self):
        if self.h and not getattr(self, "_borrowed", False):
            self.L.nlk_ctx_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise NlkError(f"rc={rc}: " + self.L.nlk_last_error(self.h).decode())

    def set_stream(self, hip_stream):
        self._chk(self.L.nlk_ctx_set_stream(self.h, hip_stream))

    def set_deterministic(self, on):
        """Bit-reproducible aggregation (slabs + ordered gather instead of float atomics)."""
        self._chk(self.L.nlk_ctx_set_deterministic(self.h, int(on)))

    def set_profiling(self, on):
        self._chk(self.L.nlk_ctx_set_profiling(self.h, int(on)))

    def timings(self):
        t = Timings()
        self._chk(self.L.nlk_ctx_get_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def sync(self):
        self._chk(self.L.nlk_sync(self.h))

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.nlk_dev_alloc(self.h, C.byref(p), nbytes))
        return p.value

    def free(self, dptr):
        self._chk(self.L.nlk_dev_free(self.h, dptr))

    def d2d(self, d_dst, d_src, nbytes):
        self._chk(self.L.nlk_d2d(self.h, d_dst, d_src, nbytes))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        d = self.alloc(arr.nbytes)
        self._chk(self.L.nlk_h2d(self.h, d, arr.ctypes.data, arr.nbytes))
        return d

    def download(self, dptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        self._chk(self.L.nlk_d2h(self.h, out.ctypes.data, dptr, out.nbytes))
        return out

    def rgb2opp(self, d_im, w, h, ch):
        self._chk(self.L.nlk_dev_rgb2opp(self.h, d_im, w, h, ch))

    def opp2rgb(self, d_im, w, h, ch):
        self._chk(self.L.nlk_dev_opp2rgb(self.h, d_im, w, h, ch))

    def warp_bicubic(self, d_out, d_im, d_flow, d_occ, w, h, ch):
        self._chk(self.L.nlk_dev_warp_bicubic(self.h, d_out, d_im, d_flow, d_occ, w, h, ch))

    def filter_frame(self, d_out, d_nisy, d_prev, d_basic, w, h, ch, sigma, params):
        self._chk(self.L.nlk_dev_filter_frame(self.h, d_out, d_nisy, d_prev, d_basic, w, h,
                                              ch, float(sigma), C.byref(params)))

    def smooth_frame(self, d_out, d_filt, d_prev, d_basic, w, h, ch, sigma, params):
        self._chk(self.L.nlk_dev_smooth_frame(self.h, d_out, d_filt, d_prev, d_basic, w, h,
                                              ch, float(sigma), C.byref(params)))

    def tvl1_flow(self, d_flow, d_i0, d_i1, w, h, params):
        """Flow I0 -> I1 into d_flow (w*h interleaved pairs); returns the iteration count."""
        it = C.c_int()
        self._chk(self.L.nlk_dev_tvl1_flow(self.h, d_flow, d_i0, d_i1, w, h, C.byref(params),
                                           C.byref(it)))
        return it.value

    def gray(self, d_gray, d_im, w, h, ch):
        self._chk(self.L.nlk_dev_gray(self.h, d_gray, d_im, w, h, ch))

    def occlusion_mask(self, d_mask, d_flow, w, h, th):
        self._chk(self.L.nlk_dev_occlusion_mask(self.h, d_mask, d_flow, w, h, float(th)))

    def image_dct(self, d_img, w, h, ch, inverse=False):
        self._chk(self.L.nlk_dev_image_dct(self.h, d_img, w, h, ch, int(inverse)))

    def copy_block(self, d_dst, dw, d_src, sw, ch, bw, bh):
        self._chk(self.L.nlk_dev_copy_block(self.h, d_dst, dw, d_src, sw, ch, bw, bh))

    def frame_accumulate(self, d_acc, d_cur, d_prev, d_basic, w, h, ch, sigma, params, oy,
                         ngy, smoother=False):
        self._chk(self.L.nlk_dev_frame_accumulate(self.h, d_acc, d_cur, d_prev, d_basic, w, h,
                                                  ch, float(sigma), C.byref(params), oy, ngy,
                                                  int(smoother)))

    def strip_match(self, d_marks, d_cur, d_prev, d_basic, w, h, ch, sigma, params, oy, ngy,
                    smoother=False):
        """Phase 1 on a strip; returns the marking reach R for mask_commit."""
        r = C.c_int()
        self._chk(self.L.nlk_dev_strip_match(self.h, d_cur, d_prev, d_basic, w, h, ch, float(sigma),
                                             C.byref(params), oy, ngy, int(smoother), d_marks,
                                             C.byref(r)))
        return r.value

    def strip_match_rows(self, d_marks, d_cur, d_prev, d_basic, w, h, ch, sigma, params, oy, ngy, r0, rows,
                         smoother=False):
        """Phase 1 for the strip's target rows [r0, r0 + rows) only (marks / records at their place)."""
        r = C.c_int()
        self._chk(self.L.nlk_dev_strip_match_rows(self.h, d_cur, d_prev, d_basic, w, h, ch, float(sigma),
                                                  C.byref(params), oy, ngy, int(smoother), r0, rows, d_marks,
                                                  C.byref(r)))
        return r.value

    def strip_match_part(self, d_marks, d_cur, d_prev, d_basic, w, h, ch, sigma, params, oy, ngy, r0, rows, lay,
                         smoother=False):
        """strip_match_rows that lays out only the pixel rows lay = (lay0, lay1, v0, v1) (include/nlk_hip.h)."""
        r = C.c_int()
        self._chk(self.L.nlk_dev_strip_match_part(self.h, d_cur, d_prev, d_basic, w, h, ch, float(sigma),
                                                  C.byref(params), oy, ngy, int(smoother), r0, rows,
                                                  int(lay[0]), int(lay[1]), int(lay[2]), int(lay[3]), d_marks, C.byref(r)))
        return r.value

    def mask_commit(self, d_marks, ngx, ngy, reach, d_active):
        self._chk(self.L.nlk_dev_mask_commit(self.h, d_marks, ngx, ngy, reach, d_active))

    def strip_group(self, d_acc, d_active):
        self._chk(self.L.nlk_dev_strip_group(self.h, d_acc, d_active))

    def strip_commit_group(self, d_acc, d_marks, ngx, ngy, reach, gy0, d_active):
        """mask_commit over the whole grid + strip_group of the strip starting at grid row gy0, as one call (the
        replay inside the group kernel's launch where it can: include/nlk_hip.h)."""
        self._chk(self.L.nlk_dev_strip_commit_group(self.h, d_acc, d_marks, ngx, ngy, reach, gy0, d_active))

    def frame_normalize(self, d_out, d_acc, d_cur, w, h, ch, y0, y1):
        self._chk(self.L.nlk_dev_frame_normalize(self.h, d_out, d_acc, d_cur, w, h, ch, y0, y1))

    def read_records(self):
        n, k, g = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.nlk_ctx_read_records(self.h, C.byref(n), C.byref(k), C.byref(g),
                                              None, None, None, None, None, None))
        n, k, g = n.value, k.value, g.value
        rec = dict(active=np.zeros(n, np.uint8), nsel=np.zeros(n, np.int32),
                   np0=np.zeros(n, np.int32), nagg=np.zeros(n, np.int32),
                   topk=np.zeros((n, k), np.uint32), gcoords=np.zeros((n, g), np.uint32))
        self._chk(self.L.nlk_ctx_read_records(
            self.h, None, None, None, rec["active"].ctypes.data, rec["nsel"].ctypes.data,
            rec["np0"].ctypes.data, rec["nagg"].ctypes.data, rec["topk"].ctypes.data,
            rec["gcoords"].ctypes.data))
        return rec


class Strips:
    """nlk_strips wrapper (include/nlk_hip.h, csrc/strips.hip): one frame over `world` row strips, driven from C.
    devices = one HIP device index (this process holds rank `rank0` of the world: RCCL between the processes,
    call rccl_init) or `world` of them (every strip in this process, device copies; indices may repeat)."""
    PHASES = ("exchange_prev", "match", "marks", "commit", "group", "exchange_acc", "normalize")

    def __init__(self, devices, rank0, world, w, h, ch, sigma, params, smoother=False, have_prev=True):
        self.L = hip()
        self.h = C.c_void_p()
        self.w, self.hh, self.ch, self.world, self.nlocal = w, h, ch, world, len(devices)
        dev = (C.c_int * len(devices))(*devices)
        rc = self.L.nlk_strips_create(C.byref(self.h), len(devices), dev, rank0, world, w, h, ch, float(sigma),
                                      C.byref(params), int(smoother), int(have_prev))
        if rc:
            raise NlkError(f"rc={rc}: " + self.L.nlk_last_error(None).decode())

    def _chk(self, rc):
        if rc:
            raise NlkError(f"rc={rc}: " + self.L.nlk_strips_last_error(self.h).decode() + " / " + self.L.nlk_last_error(None).decode())

    def close(self):
        if self.h:
            self.L.nlk_strips_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = hip().nlk_rccl_unique_id(buf)
        if rc:
            raise NlkError(hip().nlk_last_error(None).decode())
        return bytes(buf.raw)

    def rccl_init(self, id128):
        self._chk(self.L.nlk_strips_rccl_init(self.h, C.c_char_p(id128)))

    def transport(self):
        return self.L.nlk_strips_transport(self.h).decode()

    def load(self, local, d_cur_full, d_prev_full):
        self._chk(self.L.nlk_strips_load(self.h, local, d_cur_full, d_prev_full))

    def set_options(self, overlap=False, timing=False, graph=False):
        self._chk(self.L.nlk_strips_set_options(self.h, int(overlap), int(timing), int(graph)))

    def set_dry_run(self, on=True):
        self._chk(self.L.nlk_strips_set_dry_run(self.h, int(on)))

    def step(self):
        self._chk(self.L.nlk_strips_step(self.h))

    def sync(self):
        self._chk(self.L.nlk_strips_sync(self.h))

    def geometry(self, local=0):
        g = (C.c_int * 6)()
        self._chk(self.L.nlk_strips_geometry(self.h, local, g))
        return dict(zip(("gy0", "gy1", "Y0", "Y1", "own0", "own1"), g))

    def own_rows(self, local=0):
        """(y0, y1, device pointer to the output rows, pointer to the whole-grid decisions)"""
        y0, y1, rows, act = C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.nlk_strips_own_rows(self.h, local, C.byref(y0), C.byref(y1), C.byref(rows), None, C.byref(act)))
        return y0.value, y1.value, rows.value, act.value

    def download_rows(self, local=0):
        y0, y1, rows, _ = self.own_rows(local)
        out = np.empty((y1 - y0, self.w, self.ch), np.float32)
        c = self.L.nlk_strips_ctx(self.h, local)
        rc = self.L.nlk_d2h(c, out.ctypes.data, rows, out.nbytes)
        if rc:
            raise NlkError(self.L.nlk_last_error(c).decode())
        return y0, y1, out

    def stats(self):
        ph, us, g = (C.c_float * 7)(), C.c_float(), C.c_int()
        self._chk(self.L.nlk_strips_stats(self.h, ph, C.byref(us), C.byref(g)))
        return dict(zip(self.PHASES, (round(float(v), 4) for v in ph))), float(us.value), bool(g.value)
