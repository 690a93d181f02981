#!/bin/bash
# The F1 (TV-L1) counter passes of tools/profile_round.sh alone: tools/profile_pmc_f1.sh <tag>
set -u
TAG=${1:-r05}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout 300"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  $T rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_F1/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --workload F1 > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_F1 > $OUT/pmc_F1_summary.txt 2>&1
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
