#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dct_forms or stagewise or integer_records or deterministic or mask_replay or edge_cases or randomised" > gpurun_out/r06a/pytest_subset.log 2>&1
tail -3 gpurun_out/r06a/pytest_subset.log
bash tools/ab_bench.sh > gpurun_out/r06a/ab_bench.txt 2>&1
cat gpurun_out/r06a/ab_bench.txt
bash tools/ab_pmc_lds.sh r06a_lds > gpurun_out/r06a/ab_pmc_lds.txt 2>&1
cat gpurun_out/r06a/ab_pmc_lds.txt
