"""profiles/<tag>_traffic.json from the PMC summaries of tools/profile_round.sh: FETCH_SIZE + WRITE_SIZE
per launch of the filter kernels (rocprofv3 reports KiB; x 1024), with the hash of the kernel sources
they were measured on. bench.py reports `roofline.traffic` from this table only while that hash is
still the hash of the sources it runs.
   python tools/make_traffic.py <tag>      (reads gpurun_out/<tag>/pmc_{C2,C3}_summary.txt)"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path):
    """{entry: ({counter: mean}, launches)}; entry = 'instance grid=G lds=L' (tools/pmc_summary.py)"""
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = [{}, 0]
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
            if m and cur:
                out[cur][0][m.group(1)] = float(m.group(3))
                out[cur][1] = max(out[cur][1], int(m.group(2)))
    return out


def git_head():
    try:   # (no .git on the GPU box: the table made there names the sources by their hash only)
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return os.environ.get("NLK_GIT_HEAD", "unknown (made on the GPU box: see kernel_sources_sha256)")


def main(tag):
    from bench import kernel_sources_sha
    res = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh {tag}), "
                     "bench.py --steps 3 --warmup 1 --no-cpu; mean per launch",
           "git_head": git_head(),
           "kernel_sources_sha256": kernel_sources_sha(),
           "note": "raw FETCH_SIZE + WRITE_SIZE (KiB) x 1024. FETCH_SIZE counts the L2's fabric-side read requests "
                   "(Infinity-Cache hits included) and reports half the bytes of wide (16 B per lane) coalesced streaming "
                   "reads on gfx950 (MI355X_MICROARCH.md): `traffic_bytes_fetch_x2` doubles it, the upper bound for "
                   "kernels whose reads are such loads; WRITE_SIZE is exact for 16-B stores and float atomics.",
           "workloads": {}}
    for w in ("C2", "C3", "C5", "C1L", "F1"):
        p = os.path.join(ROOT, "gpurun_out", tag, f"pmc_{w}_summary.txt")
        if not os.path.exists(p):
            continue
        ks = {}
        for name, (c, nlaunch) in parse(p).items():
            if "FETCH_SIZE" not in c or not name.startswith("k_"):
                continue
            f, wr = c["FETCH_SIZE"] * 1024, c.get("WRITE_SIZE", 0.0) * 1024
            inst = name.split(" grid=")[0]   # full instance name, e.g. "k_bm_topk<8, 3, 2>"
            ent = {"instance": inst, "launch_shape": name[len(inst) + 1:], "launches_per_pass": nlaunch,
                   "fetch_bytes_raw": f, "write_bytes": wr, "traffic_bytes": f + wr,
                   "traffic_bytes_fetch_x2": 2 * f + wr}
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES",
                      "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                      "SQ_WAIT_INST_ANY", "TCC_EA0_ATOMIC_sum"):
                if k in c:
                    ent[k] = c[k]
            # one entry per INSTANCE: the launch shape with the most dispatches (the steady-state frames of the
            # bench; the one spatial call that builds the previous frame launches other instances / shapes)
            # (F1: the flow's block kernel runs one shape per pyramid level: all kept, keyed by shape)
            if w == "F1":
                ks[name] = ent
            elif inst not in ks or nlaunch > ks[inst]["launches_per_pass"]:
                ks[inst] = ent
        res["workloads"][w] = ks
    path = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
    json.dump(res, open(path, "w"), indent=1)
    print(path, {w: sorted(k) for w, k in res["workloads"].items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
