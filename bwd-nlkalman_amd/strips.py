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
        """phases = (match, commit, group) callbacks selects the exact mode:
        match(marks[int64, ngy_l*ngx], cur, prev, oy, ngy_l) -> reach R;
        commit(marks_full[int64], ngx, ngy, R, active_full[uint8]);
        group(acc, active_slice[uint8], ...) (state of the last match)."""
        self.rank, self.world, self.w, self.h, self.ch = rank, world, w, h, ch
        self.step_px = psz // 2
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

    def _exchange(self, sends, recvs):
        if self.stage_host:
            sends = [(t.cpu(), peer) for t, peer in sends]
            dev_recvs, recvs = recvs, [(torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
        ops = [dist.P2POp(dist.isend, t, peer) for t, peer in sends]
        ops += [dist.P2POp(dist.irecv, t, peer) for t, peer in recvs]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.stage_host:
            for (d, _), (hbuf, _) in zip(dev_recvs, recvs):
                d.copy_(hbuf)

    def _all_gather(self, flat, outs, t):
        """t of every rank -> flat (= the views `outs`, rank-major)."""
        if not self.stage_host:
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(flat, t)   # RCCL: straight into the flat buffer
            else:
                dist.all_gather(outs, t)
            return
        houts = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(houts, t.cpu())
        flat.copy_(torch.cat(houts))

    def step(self):
        p, plan, w, ch = self.p, self.plan, self.w, self.ch
        l = self._l
        # (1) halo rows of the previous denoised frame: row ranges of `prev` are contiguous, so they are
        # sent from and received into place
        sends, recvs = [], []
        if self.up is not None:
            n_up = plan[self.up]["Y1"] - p["own0"]          # my rows the upper rank reads
            sends.append((self.prev[l(p["own0"]):l(p["own0"]) + n_up], self.up))
            recvs.append((self.prev[:l(p["own0"])], self.up))
        if self.dn is not None:
            n_dn = p["own1"] - plan[self.dn]["Y0"]          # my rows the lower rank reads
            sends.append((self.prev[l(p["own1"]) - n_dn:l(p["own1"])], self.dn))
            recvs.append((self.prev[l(p["own1"]):], self.dn))
        self._exchange(sends, recvs)
        # (2) kernels on the strip
        self.acc.zero_()
        oy, ngy_l = p["gy0"] * self.step_px - p["Y0"], p["gy1"] - p["gy0"]
        if self.phases is None:
            self.accumulate(self.acc, self.cur, self.prev, oy, ngy_l)
        else:
            match, commit, group = self.phases
            n_l = ngy_l * self.ngx
            reach = match(self.marks_pad[:n_l], self.cur, self.prev, oy, ngy_l)
            if self.world > 1:
                self._all_gather(self.marks_flat, self.marks_all, self.marks_pad)
                torch.index_select(self.marks_flat, 0, self.compact, out=self.marks_full)
            else:
                self.marks_full.copy_(self.marks_pad[:n_l])
            commit(self.marks_full, self.ngx, self.ngy, reach, self.active_full)
            group(self.acc, self.active_full[p["gy0"] * self.ngx:p["gy1"] * self.ngx])
        # (3) accumulator rows written outside the own rows go to their owner
        sends, recvs, at, ab = [], [], None, None
        if self.up is not None:
            sends.append((self.acc[:, :l(p["own0"])].contiguous(), self.up))
            at = torch.empty((ch + 1, plan[self.up]["Y1"] - p["own0"], w), dtype=torch.float32,
                             device=self.cur.device)
            recvs.append((at, self.up))
        if self.dn is not None:
            sends.append((self.acc[:, l(p["own1"]):].contiguous(), self.dn))
            ab = torch.empty((ch + 1, p["own1"] - plan[self.dn]["Y0"], w), dtype=torch.float32,
                             device=self.cur.device)
            recvs.append((ab, self.dn))
        self._exchange(sends, recvs)
        if at is not None:
            self.acc[:, l(p["own0"]):l(p["own0"]) + at.shape[1]] += at
        if ab is not None:
            self.acc[:, l(p["own1"]) - ab.shape[1]:l(p["own1"])] += ab
        # (4) normalise the own rows
        self.normalize(self.out, self.acc, self.cur, l(p["own0"]), l(p["own1"]))

    def own_rows(self):
        p = self.p
        return p["own0"], p["own1"], self.out[self._l(p["own0"]):self._l(p["own1"])]
