#!/bin/bash
# The LDS / wait counters of the C2 launch for every library under bwd-nlkalman_amd/ab/*.so (one --pmc pass each):
#   tools/ab_pmc_lds.sh <tag> [bench args...]   -> gpurun_out/<tag>/<lib>/ + a one-line summary per library
set -u
TAG=${1:-ablds}; shift || true
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cp $ROOT/bwd-nlkalman_amd/libnlk_hip.so /tmp/libnlk_hip_current.so
cd /tmp && export TMPDIR=/tmp
for L in $ROOT/bwd-nlkalman_amd/ab/*.so; do
  n=$(basename $L .so)
  cp $L $ROOT/bwd-nlkalman_amd/libnlk_hip.so
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/$n/p1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extras "$@" > $OUT/$n.log 2>&1
  echo "== $n"
  python3 $ROOT/tools/pmc_summary.py $OUT/$n 2>&1 | grep -A12 "k_group8m" | head -14
done
cp /tmp/libnlk_hip_current.so $ROOT/bwd-nlkalman_amd/libnlk_hip.so
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
