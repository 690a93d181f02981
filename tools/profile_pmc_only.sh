#!/bin/bash
# The rocprofv3 part of tools/profile_round.sh alone (kernel statistics + PMC passes of C2 and C3): tools/profile_pmc_only.sh <tag>
set -u
TAG=${1:-r06}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout 300"
for W in C2 C3 C5 C1L; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o s -- python3 $ROOT/bench.py --no-cpu --no-extras --workload $W > $OUT/bench_${W}_under_rocprof.json 2>/dev/null
  i=0
  for SET in \
   "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
   "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
   "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 GRBM_GUI_ACTIVE" \
   "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"; do
    i=$((i+1))
    $T rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$W/p$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extras --workload $W > /dev/null 2>&1
  done
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_$W > $OUT/pmc_${W}_summary.txt 2>&1
done
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
