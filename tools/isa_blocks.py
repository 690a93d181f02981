"""Per basic block instruction mix of one kernel in a hipcc -S listing (which loop carries what):
   python tools/isa_blocks.py /tmp/g8.s _Z9k_group8mILi3ELb0ELb1E [min_instructions]"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
on = False
blocks = []
cur = None
for line in open(path):
    if not on:
        if line.startswith(prefix) and line.rstrip().endswith(":") or (line.startswith(prefix) and ":" in line):
            on = True
            cur = {"name": "entry", "note": "", "n": {}}
            blocks.append(cur)
        continue
    s = line.strip()
    if s.startswith(".LBB") or re.match(r"^\.LBB\d+_\d+:", s):
        note = s.split(";", 1)[1].strip() if ";" in s else ""
        cur = {"name": s.split(":")[0], "note": note, "n": {}}
        blocks.append(cur)
        continue
    if s.startswith(";") and cur is not None and ("Loop" in s or "Depth" in s):
        cur["note"] += " " + s.lstrip("; ")
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    if op == "s_endpgm":
        break
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith("scratch_"): k = "scratch"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): k = "vmem"
    elif op == "s_nop": k = "nop"
    elif op.startswith("s_waitcnt"): k = "wait"
    elif op.startswith("v_permlane") or "dpp" in s: k = "xlane"
    elif op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): k = "rdlane"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cur["n"][k] = cur["n"].get(k, 0) + 1
for b in blocks:
    tot = sum(b["n"].values())
    if tot >= minn:
        print(f"{b['name']:12s} {tot:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(b["n"].items())) + "   | " + b["note"][:70])
