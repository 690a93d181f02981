#!/bin/bash
# Timeline (kernels + memory copies) of the host-pointer frame call: rocprofv3 traces of tools/api_wall.py
#   tools/api_trace.sh <tag>   -> gpurun_out/<tag>/timeline.txt (the last call's operations, ms from its first copy)
TAG=${1:-api_trace}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
NLK_API_WALL_ONLY=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o t -- python3 $GRAFT_REPO_ROOT/tools/api_wall.py > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
ops = []
for f in glob.glob(os.path.join(out, "tr", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(os.path.join(out, "tr", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ops.sort()
# the last frame call: operations after the last gap of > 5 ms
cut = 0
for i in range(1, len(ops)):
    if ops[i][0] - ops[i - 1][1] > 3_000_000:
        cut = i
ops = ops[cut:]
t0 = ops[0][0]
with open(os.path.join(out, "timeline.txt"), "w") as fo:
    for a, b, n in ops:
        fo.write(f"{(a - t0) / 1e6:8.3f} {(b - t0) / 1e6:8.3f}  {(b - a) / 1e6:7.3f}  {n}\n")
print(open(os.path.join(out, "timeline.txt")).read()[-6000:])
PY
find $OUT -name "*.csv" -size +4M -delete; find $OUT -name "*.db" -delete
