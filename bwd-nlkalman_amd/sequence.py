"""In-process sequence driver: the recursion of
scripts/nlkalman-seq.sh — per frame `tvl1flow` -> occlusion mask -> `nlkalman-flt`
(iteration 1) -> `nlkalman-flt` (iteration 2), then backwards `tvl1flow` -> mask ->
`nlkalman-smo` — with every frame, flow and mask resident in HBM and one device context,
instead of four processes and ~10 float-TIFF round trips per frame.

Forward (reference: scripts/nlkalman-seq.sh:30-115):
    frame 0:  flt1_0 = FLT1(noisy_0),  flt2_0 = FLT2(noisy_0, basic = flt1_0)
    frame t:  flow_t = TVL1(noisy_t -> flt2_{t-1});  occ_t = |div flow_t| > TH
              flt1_t = FLT1(noisy_t, warp(flt1_{t-1}, flow_t, occ_t))
              flt2_t = FLT2(noisy_t, warp(flt2_{t-1}, flow_t, occ_t), basic = flt1_t)
Backward (reference: :117-150):
    smo_T = flt2_T;  smo_t = SMO1(flt2_t, warp(smo_{t+1}, TVL1(flt2_t -> smo_{t+1}), occ))

Frames are kept in the opponent colour space the filters work in (src/main-flt.c:335-343
converts on read and back on write); the flow sees the luminance of the RGB frames like
the tool does. Compared with the chain of processes this skips one opp->rgb->opp rounding
of the previous outputs per frame (~1e-5 on the 0..255 scale); nothing else differs.

Host side only: every array operation is a call of the C-ABI (include/nlk_hip.h).
"""
import importlib

_pkg = None


def _p():
    global _pkg
    if _pkg is None:
        _pkg = importlib.import_module(__name__.rsplit(".", 1)[0])
    return _pkg


class SequenceFilter:
    """`ctx` = bwd-nlkalman_amd.Context. Flow parameters `of_*` are what the scripts pass to
    tvl1flow (lambda = "DW", finest scale, occlusion threshold: nlkalman-seq.sh:47-52); the defaults
    are the script's own OPM default "1 0.25 0.75 1 0.25 0.75" (nlkalman-seq.sh:12), as in host/main_seq.c
    (nlkalman-seq-gt.sh uses DW = 0.40)."""

    def __init__(self, ctx, w, h, ch, sigma, f1=None, f2=None, s1=None, of_lambda=0.25, of_fscale=1,
                 occ_th=0.75, keep_history=True):
        pkg = _p()
        self.ctx, self.w, self.h, self.ch, self.sigma = ctx, w, h, ch, float(sigma)
        self.f1 = f1 or pkg.default_params(sigma, pkg.FLT1)
        self.f2 = f2 or pkg.default_params(sigma, pkg.FLT2)
        self.s1 = s1 or pkg.default_params(sigma, pkg.SMO1)
        self.of = pkg.tvl1_params(w, h, lam=of_lambda, fscale=of_fscale)
        self.occ_th = float(occ_th)
        self.nbytes = w * h * ch * 4
        a = ctx.alloc
        self.d_noisy, self.d_rgb, self.d_warp = a(self.nbytes), a(self.nbytes), a(self.nbytes)
        self.d_g0, self.d_g1, self.d_occ = a(w * h * 4), a(w * h * 4), a(w * h * 4)
        self.d_flow = a(w * h * 8)
        self.flt1, self.flt2 = None, None   # previous outputs (opponent space, device)
        self.history = [] if keep_history else None  # flt2 of every frame, for smooth()
        self._pool = []                      # released frame buffers (no hipMalloc / hipFree per frame)
        self.t = 0
        self.flow_iterations = []
        self.stage_s = None   # set to {} to have push() synchronise after each stage and add up its wall times (bench.py S1)

    def _flow_and_mask(self, d_from_rgb, d_to_opp):
        """flow from frame `d_from_rgb` (RGB) to the frame whose opponent image is `d_to_opp`."""
        c, w, h, ch = self.ctx, self.w, self.h, self.ch
        c.gray(self.d_g0, d_from_rgb, w, h, ch)
        c.d2d(self.d_rgb, d_to_opp, self.nbytes)
        c.opp2rgb(self.d_rgb, w, h, ch)
        c.gray(self.d_g1, self.d_rgb, w, h, ch)
        self.flow_iterations.append(c.tvl1_flow(self.d_flow, self.d_g0, self.d_g1, w, h, self.of))
        c.occlusion_mask(self.d_occ, self.d_flow, w, h, self.occ_th)

    def push(self, d_noisy_rgb):
        """Next noisy frame (device pointer, HWC RGB or gray, not modified). Afterwards
        self.flt1 / self.flt2 hold its two estimates (opponent space)."""
        c, w, h, ch, sg = self.ctx, self.w, self.h, self.ch, self.sigma
        c.d2d(self.d_noisy, d_noisy_rgb, self.nbytes)
        c.rgb2opp(self.d_noisy, w, h, ch)
        n1, n2 = self._frame(), self._frame()
        if self.t == 0:
            c.filter_frame(n1, self.d_noisy, None, None, w, h, ch, sg, self.f1)
            c.filter_frame(n2, self.d_noisy, None, n1, w, h, ch, sg, self.f2)
        else:
            mark = self._stage_mark
            mark(None)
            self._flow_and_mask(d_noisy_rgb, self.flt2)
            mark("flow_and_mask")
            c.warp_bicubic(self.d_warp, self.flt1, self.d_flow, self.d_occ, w, h, ch)
            c.filter_frame(n1, self.d_noisy, self.d_warp, None, w, h, ch, sg, self.f1)
            mark("warp_flt1")
            c.warp_bicubic(self.d_warp, self.flt2, self.d_flow, self.d_occ, w, h, ch)
            c.filter_frame(n2, self.d_noisy, self.d_warp, n1, w, h, ch, sg, self.f2)
            mark("warp_flt2")
        if self.flt1:
            self._pool.append(self.flt1)
        if self.flt2 and self.history is None:
            self._pool.append(self.flt2)
        self.flt1, self.flt2 = n1, n2
        if self.history is not None:
            self.history.append(n2)
        self.t += 1

    def _frame(self):
        return self._pool.pop() if self._pool else self.ctx.alloc(self.nbytes)

    def _stage_mark(self, name):
        """(profiling only: self.stage_s is a dict) wall time since the last mark, after a device sync, added to `name`"""
        if self.stage_s is None:
            return
        import time
        self.ctx.sync()
        now = time.perf_counter()
        if name is not None:
            self.stage_s[name] = self.stage_s.get(name, 0.0) + now - self._stage_t
        self._stage_t = now

    def smooth(self, of_lambda=None, of_fscale=None, occ_th=None):
        """Backward pass over the kept flt2 frames; returns the list of smoothed frames
        (device pointers, opponent space; the last one is flt2 of the last frame)."""
        if self.history is None:
            raise RuntimeError("SequenceFilter(keep_history=False) kept no frames to smooth")
        pkg = _p()
        c, w, h, ch = self.ctx, self.w, self.h, self.ch
        of_fwd, th_fwd = self.of, self.occ_th
        self.of = pkg.tvl1_params(w, h, lam=of_lambda if of_lambda is not None else self.of.lam,
                                  fscale=of_fscale if of_fscale is not None else self.of.fscale)
        if occ_th is not None:
            self.occ_th = float(occ_th)
        out = [None] * len(self.history)
        out[-1] = self.history[-1]
        for t in range(len(self.history) - 2, -1, -1):
            # the flow tool is given flt2_t as an RGB file: convert a copy
            c.d2d(self.d_noisy, self.history[t], self.nbytes)
            c.opp2rgb(self.d_noisy, w, h, ch)
            self._flow_and_mask(self.d_noisy, out[t + 1])
            c.warp_bicubic(self.d_warp, out[t + 1], self.d_flow, self.d_occ, w, h, ch)
            out[t] = c.alloc(self.nbytes)
            c.smooth_frame(out[t], self.history[t], self.d_warp, None, w, h, ch, self.sigma, self.s1)
        self.of, self.occ_th = of_fwd, th_fwd
        return out

    def download_rgb(self, d_opp):
        """Host RGB copy of a resident opponent-space frame."""
        c = self.ctx
        c.d2d(self.d_rgb, d_opp, self.nbytes)
        c.opp2rgb(self.d_rgb, self.w, self.h, self.ch)
        return c.download(self.d_rgb, (self.h, self.w, self.ch))
