#!/bin/bash
# One --pmc pass of bench.py with the given counters:  tools/pmc_one.sh <tag> "<COUNTERS...>" [bench args...]
set -u
TAG=$1; SET=$2; shift 2
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extras "$@" > $OUT/log.txt 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT | grep -A12 "k_group8m\|k_bm_topk" | grep -v "^--"
find $OUT -name "*.csv" -size +6M -delete; find $OUT -name "*.db" -delete
