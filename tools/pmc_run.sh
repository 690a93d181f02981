#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py in separate passes (one counter
# set per pass, kernel-trace only, as the MI355X guide prescribes).
#   tools/pmc_run.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/p$i.log 2>&1
  echo "pass $i: $(grep -c . $OUT/p$i.log) log lines"
done
find $OUT -name "*counter_collection.csv" | head
