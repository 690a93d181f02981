#!/bin/bash
# SQ instruction / wait / LDS counters of one bench workload, two rocprofv3 --pmc passes
# (kernel-trace only), summarised per kernel.   tools/pmc_sq.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A30 "k_group" $OUT/summary.txt | head -80
find $OUT -name "*.csv" -size +4M -delete
