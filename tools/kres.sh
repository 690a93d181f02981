#!/bin/bash
# Registers, spills, scratch of every kernel of a translation unit (compiler view):  tools/kres.sh csrc/tu_group8.hip [extra flags]
cd "$(dirname "$0")/../bwd-nlkalman_amd"
f=$1; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize -I../include --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c -o /tmp/kres.o "$f" 2>&1 |
  python3 -c '
import re, sys
cur = None
keep = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]")
for line in sys.stdin:
    m = re.search(r"remark: +([^:]+): +(\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur: print(cur)
        cur = v + ":"
    elif k in keep:
        cur += " %s=%s" % (k.split(" [")[0], v)
if cur: print(cur)
'
