"""Row-strip sharding of one frame over N ranks (one process per GPU).

The frame's patch-grid rows are split into N contiguous strips. Rank r keeps
its pixel rows plus the search halo (`halo` = largest search radius: a target
reads candidates up to `halo` rows away and writes group members up to
`halo + psz - 1` rows away, reference: src/nlkalman.c:637-639, 916-927).
Per frame each rank

  1. receives the halo rows of the PREVIOUS denoised frame from its neighbours
     (the rows it does not own; in a video loop the previous output is resident
     strip-wise, so this is the only input traffic between GPUs),
  2. runs the kernels on its strip (nlk_dev_frame_accumulate),
  3. sends the accumulator rows it wrote outside its own rows to their owner
     and adds what it receives,
  4. normalises its own rows (nlk_dev_frame_normalize).

Exchanges are neighbour send/recv pairs (torch.distributed P2P: RCCL over
xGMI with the "nccl" backend, TCP with "gloo" in the CPU tests). The
processed-mask is replayed per strip, exactly like the reference's static
OpenMP split of the grid rows (reference: src/nlkalman.c:586): targets near a
seam may be skipped differently than in the serial order (measured ~0.001 dB).

EXACT mode (`phases=` given): step 2 is split in three. Every rank matches its
strip and produces one 64-bit mark word per target; the mark words of all
strips are all-gathered (the one real collective of the path, ~1 MB at 1080p);
every rank replays the raster-order mask over the WHOLE patch grid (integer
work, 0.2 ms) and processes its strip with its slice of the result. The output
then equals the single-GPU / serial order exactly, for any number of ranks.

The compute callbacks are injected so that this module never imports a
backend: bench.py passes the HIP C-ABI, the CPU tests pass the oracle.
"""
import torch
import torch.distributed as dist


def strip_plan(h, psz, halo, world):
    """Per rank: dict(gy0, gy1, Y0, Y1, own0, own1) in frame rows."""
    step = psz // 2
    ngy = (h - psz) // step + 1
    if ngy < world:
        raise ValueError(f"{ngy} patch-grid rows cannot be split over {world} ranks")
    plan = []
    for r in range(world):
        gy0, gy1 = (ngy * r) // world, (ngy * (r + 1)) // world
        plan.append(dict(
            gy0=gy0, gy1=gy1,
            Y0=max(0, gy0 * step - halo), Y1=min(h, (gy1 - 1) * step + halo + psz),
            own0=gy0 * step if r > 0 else 0, own1=gy1 * step if r < world - 1 else h))
    for a, b in zip(plan[:-1], plan[1:]):
        # a strip may only spill into its direct neighbour's own rows
        if a["Y1"] > b["own1"] or b["Y0"] < a["own0"]:
            raise ValueError("strips thinner than the search halo: use fewer ranks")
    return plan


class StripFrame:
    """Buffers + per-frame step of one rank. `accumulate(acc, cur, prev, oy, ngy)`
    adds the strip's weighted patches into the planar accumulator tensor
    acc[(ch+1), hl, w]; `normalize(out, acc, cur, y0, y1)` writes rows [y0, y1)."""

    def __init__(self, rank, world, w, h, ch, psz, halo, device, accumulate, normalize,
                 phases=None, stage_host=False):
        """phases = (match, commit, group[, match_rows]) callbacks selects the exact mode:
        match(marks[int64, ngy_l*ngx], cur, prev, oy, ngy_l) -> reach R;
        match_rows(marks, cur, prev, oy, ngy_l, r0, rows, lay) -> R: the same for the target rows [r0, r0+rows)
        only, after laying out the pixel rows lay = (lay0, lay1, v0, v1) of the strip (include/nlk_hip.h:
        nlk_dev_strip_match_part; backends without a layout step ignore it). Optional: overlaps the matching of
        the interior rows with the previous-frame halo exchange - the own rows are laid out and the targets that
        read nothing else are matched while the halo rows travel, the halo rows are laid out and the seam targets
        matched after their arrival: nothing reads rows in flight, nothing is laid out twice;
        commit(marks_full[int64], ngx, ngy, R, active_full[uint8]);
        group(acc, active_slice[uint8], ...) (state of the last match)."""
        self.rank, self.world, self.w, self.h, self.ch = rank, world, w, h, ch
        self.step_px = psz // 2
        self.psz = psz
        self.plan = strip_plan(h, psz, halo, world)
        p = self.plan[rank]
        self.p = p
        self.hl = p["Y1"] - p["Y0"]
        f32 = dict(dtype=torch.float32, device=device)
        self.cur = torch.zeros((self.hl, w, ch), **f32)
        self.prev = torch.zeros((self.hl, w, ch), **f32)
        self.out = torch.zeros((self.hl, w, ch), **f32)
        self.acc = torch.zeros((ch + 1, self.hl, w), **f32)
        self.accumulate, self.normalize = accumulate, normalize
        self.phases = phases
        # stage_host: move exchanged tensors through host memory (lets the "gloo" backend
        # drive GPU-resident strips in tests; RCCL exchanges device memory directly)
        self.stage_host = stage_host
        self.ngx = (w - psz) // self.step_px + 1
        self.ngy = (h - psz) // self.step_px + 1
        self.rows = [q["gy1"] - q["gy0"] for q in self.plan]
        if phases is not None:
            mx = max(self.rows) * self.ngx
            self.marks_pad = torch.zeros(mx, dtype=torch.int64, device=device)
            # every rank's (padded) mark words land in one flat buffer; one gather compacts them to
            # the whole grid (instead of a copy per rank)
            self.marks_flat = torch.zeros(world * mx, dtype=torch.int64, device=device)
            self.marks_all = list(self.marks_flat.view(world, mx).unbind(0))
            self.compact = torch.cat([torch.arange(nr * self.ngx, dtype=torch.int64) + r * mx
                                      for r, nr in enumerate(self.rows)]).to(device)
            self.marks_full = torch.zeros(self.ngy * self.ngx, dtype=torch.int64, device=device)
            self.active_full = torch.zeros(self.ngy * self.ngx, dtype=torch.uint8, device=device)
        self.up = rank - 1 if rank > 0 else None
        self.dn = rank + 1 if rank < world - 1 else None
        # ---- every exchange buffer is allocated once (a frame step allocates nothing):
        # accumulator halos are packed into / received in contiguous (ch+1, rows, w) buffers
        l, plan = self._l, self.plan
        self.n_up = plan[self.up]["Y1"] - p["own0"] if self.up is not None else 0   # my rows the upper rank reads / writes
        self.n_dn = p["own1"] - plan[self.dn]["Y0"] if self.dn is not None else 0
        self.h_top, self.h_bot = l(p["own0"]), self.hl - l(p["own1"])               # my halo rows above / below
        f32 = dict(dtype=torch.float32, device=device)
        self.snd_top = torch.zeros((ch + 1, self.h_top, w), **f32) if self.up is not None else None
        self.snd_bot = torch.zeros((ch + 1, self.h_bot, w), **f32) if self.dn is not None else None
        self.rcv_top = torch.zeros((ch + 1, self.n_up, w), **f32) if self.up is not None else None
        self.rcv_bot = torch.zeros((ch + 1, self.n_dn, w), **f32) if self.dn is not None else None
        if stage_host:  # (gloo tests: pinned-size host mirrors, also allocated once)
            self._host = {}
        # target rows of the strip (strip-local indices [i0, i1)) whose search windows and candidate patches
        # lie inside this rank's OWN rows of the previous frame: they can be matched before the halo arrives
        gy0, gy1 = p["gy0"], p["gy1"]
        lo = gy0 if self.up is None else -(-(p["own0"] + halo) // self.step_px)                  # ceil
        hi = gy1 if self.dn is None else (p["own1"] - halo - psz) // self.step_px + 1
        lo, hi = max(lo, gy0), min(hi, gy1)
        self.interior = (lo - gy0, max(hi, lo) - gy0)
        # per-phase wall times of the last step (seconds; `timers=True` synchronises the device
        # after every phase, for diagnosis only: bench.py --phase-times)
        self.timers = False
        self.phase_s = {}

    # ---- helpers in strip-local row coordinates
    def _l(self, y):
        return y - self.p["Y0"]

    def load(self, cur_full, prev_own_full):
        """cur: strip + halo straight from the host frame; prev: OWN rows only
        (its halo arrives by exchange in step())."""
        p = self.p
        self.cur.copy_(cur_full[p["Y0"]:p["Y1"]])
        self.prev.zero_()
        self.prev[self._l(p["own0"]):self._l(p["own1"])] = prev_own_full[p["own0"]:p["own1"]]

    def _hbuf(self, t, tag):
        b = self._host.get(tag)
        if b is None or b.shape != t.shape:
            b = self._host[tag] = torch.empty(t.shape, dtype=t.dtype)
        return b

    def _post(self, sends, recvs, tag):
        """Neighbour exchange, posted: every send / receive at once (batch_isend_irecv: one RCCL group).
        Returns what _wait needs. Tensors are contiguous row ranges or preallocated packed buffers."""
        dev_recvs = None
        if self.stage_host:
            hs = [(self._hbuf(t, (tag, "s", i)).copy_(t), peer) for i, (t, peer) in enumerate(sends)]
            hr = [(self._hbuf(t, (tag, "r", i)), peer) for i, (t, peer) in enumerate(recvs)]
            dev_recvs, sends, recvs = recvs, hs, hr
        ops = [dist.P2POp(dist.isend, t, peer) for t, peer in sends]
        ops += [dist.P2POp(dist.irecv, t, peer) for t, peer in recvs]
        return (dist.batch_isend_irecv(ops) if ops else []), dev_recvs, recvs

    def _wait(self, posted):
        reqs, dev_recvs, recvs = posted
        for req in reqs:
            req.wait()
        if dev_recvs is not None:
            for (d, _), (hbuf, _) in zip(dev_recvs, recvs):
                d.copy_(hbuf)

    def _exchange(self, sends, recvs, tag):
        self._wait(self._post(sends, recvs, tag))

    def _all_gather(self, flat, outs, t):
        """t of every rank -> flat (= the views `outs`, rank-major)."""
        if not self.stage_host:
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(flat, t)   # RCCL: straight into the flat buffer
            else:
                dist.all_gather(outs, t)
            return
        houts = [self._hbuf(o, ("ag", i)) for i, o in enumerate(outs)]
        dist.all_gather(houts, self._hbuf(t, ("ag", "s")).copy_(t))
        flat.copy_(torch.cat(houts))

    def _tick(self, name, t0):
        if self.timers:
            import time
            if self.cur.is_cuda:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            self.phase_s[name] = self.phase_s.get(name, 0.0) + (t1 - t0)
            return t1
        return t0

    def step(self):
        import time
        p, plan, w, ch = self.p, self.plan, self.w, self.ch
        l = self._l
        t0 = time.perf_counter() if self.timers else 0.0
        # (1) halo rows of the previous denoised frame: row ranges of `prev` are contiguous, so they are
        # sent from and received into place
        sends, recvs = [], []
        if self.up is not None:
            sends.append((self.prev[l(p["own0"]):l(p["own0"]) + self.n_up], self.up))
            recvs.append((self.prev[:l(p["own0"])], self.up))
        if self.dn is not None:
            sends.append((self.prev[l(p["own1"]) - self.n_dn:l(p["own1"])], self.dn))
            recvs.append((self.prev[l(p["own1"]):], self.dn))
        posted = self._post(sends, recvs, "prev")
        # (2) kernels on the strip
        self.acc.zero_()
        oy, ngy_l = p["gy0"] * self.step_px - p["Y0"], p["gy1"] - p["gy0"]
        if self.phases is None:
            self._wait(posted)
            t0 = self._tick("exchange_prev", t0)
            self.accumulate(self.acc, self.cur, self.prev, oy, ngy_l)
            t0 = self._tick("accumulate", t0)
        else:
            match, commit, group = self.phases[:3]
            match_rows = self.phases[3] if len(self.phases) > 3 else None
            n_l = ngy_l * self.ngx
            i0, i1 = self.interior
            if match_rows is not None and i1 > i0 and (self.up is not None or self.dn is not None):
                # the target rows that see only this rank's own rows of the previous frame are matched while
                # the halo rows travel; the seam rows (and a fresh layout of the strip) follow their arrival
                o0, o1, hl, ps = l(p["own0"]), l(p["own1"]), self.hl, self.psz
                # own rows (a validity-map row needs the row tests of the psz rows from it on: the last psz - 1
                # own rows wait for the halo below, the first ones are complete unless a halo lies above)
                reach = match_rows(self.marks_pad[:n_l], self.cur, self.prev, oy, ngy_l, i0, i1 - i0,
                                   (o0, o1, o0, o1 if o1 == hl else o1 - ps + 1))
                t0 = self._tick("match_interior", t0)
                self._wait(posted)
                t0 = self._tick("exchange_prev", t0)
                # halo rows above (+ the seam targets there), then below
                if o0 > 0 or i0 > 0:
                    match_rows(self.marks_pad[:n_l], self.cur, self.prev, oy, ngy_l, 0, i0, (0, o0, 0, o0))
                if o1 < hl or i1 < ngy_l:
                    match_rows(self.marks_pad[:n_l], self.cur, self.prev, oy, ngy_l, i1, ngy_l - i1,
                               (o1, hl, max(o1 - ps + 1, o0) if o1 < hl else hl, hl))
                t0 = self._tick("match_seams", t0)
            else:
                self._wait(posted)
                t0 = self._tick("exchange_prev", t0)
                reach = match(self.marks_pad[:n_l], self.cur, self.prev, oy, ngy_l)
                t0 = self._tick("match", t0)
            if self.world > 1:
                self._all_gather(self.marks_flat, self.marks_all, self.marks_pad)
                torch.index_select(self.marks_flat, 0, self.compact, out=self.marks_full)
            else:
                self.marks_full.copy_(self.marks_pad[:n_l])
            t0 = self._tick("gather_marks", t0)
            commit(self.marks_full, self.ngx, self.ngy, reach, self.active_full)
            t0 = self._tick("commit", t0)
            group(self.acc, self.active_full[p["gy0"] * self.ngx:p["gy1"] * self.ngx])
            t0 = self._tick("group", t0)
        # (3) accumulator rows written outside the own rows go to their owner: packed into the
        # preallocated buffers (one strided copy each), added on arrival
        sends, recvs = [], []
        if self.up is not None:
            self.snd_top.copy_(self.acc[:, :self.h_top])
            sends.append((self.snd_top, self.up))
            recvs.append((self.rcv_top, self.up))
        if self.dn is not None:
            self.snd_bot.copy_(self.acc[:, l(p["own1"]):])
            sends.append((self.snd_bot, self.dn))
            recvs.append((self.rcv_bot, self.dn))
        self._exchange(sends, recvs, "acc")
        if self.up is not None:
            self.acc[:, l(p["own0"]):l(p["own0"]) + self.n_up] += self.rcv_top
        if self.dn is not None:
            self.acc[:, l(p["own1"]) - self.n_dn:l(p["own1"])] += self.rcv_bot
        t0 = self._tick("exchange_acc", t0)
        # (4) normalise the own rows
        self.normalize(self.out, self.acc, self.cur, l(p["own0"]), l(p["own1"]))
        self._tick("normalize", t0)

    def own_rows(self):
        p = self.p
        return p["own0"], p["own1"], self.out[self._l(p["own0"]):self._l(p["own1"])]
