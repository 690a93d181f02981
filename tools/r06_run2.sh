#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "block_summed" > gpurun_out/r06b/pytest_block.log 2>&1
tail -15 gpurun_out/r06b/pytest_block.log
bash tools/ab_env.sh NLK_MATCH_ORDER=block C2 2 > gpurun_out/r06b/ab_match_order.txt 2>&1
cat gpurun_out/r06b/ab_match_order.txt
for o in exact block; do echo "order $o"; NLK_MATCH_ORDER=$o timeout 300 python3 tools/mode_times.py 2>/dev/null | grep -v "^$" ; done > gpurun_out/r06b/mode_times_by_order.txt 2>&1
cat gpurun_out/r06b/mode_times_by_order.txt
NLK_MATCH_BX2=0 bash tools/ab_env.sh NLK_MATCH_ORDER=block C2 1 > gpurun_out/r06b/ab_match_order_bx4.txt 2>&1
cat gpurun_out/r06b/ab_match_order_bx4.txt
