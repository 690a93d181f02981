#!/bin/bash
# Is a kernel bound by the vector-memory front end (texture addresser / L1 tag rate)? TA / TCP counters of one bench
# workload, kernel-trace + --pmc only.   tools/pmc_ta.sh <tag> [bench args...]
set -u
TAG=${1:-pmcta}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\bTA_[A-Z0-9_]*\|\bTCP_[A-Z0-9_]*\|\bTD_[A-Z0-9_]*" | sort -u > $OUT/counters.txt
i=0
for SET in \
 "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
 "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
 "TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extras "$@" > $OUT/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A16 "k_group8m.*grid=1626112" $OUT/summary.txt | head -40
