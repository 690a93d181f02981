#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
echo "exact order"; bash tools/ab_bench.sh --no-extras 2>&1 | grep "run 1"
echo "block order"; NLK_MATCH_ORDER=block bash tools/ab_bench.sh --no-extras 2>&1 | grep "run 1"
