#!/bin/bash
# Soak at the final binaries, in ONE gpurun call: full-size random configurations against the serial oracle in parallel
# processes (tools/soak_fullsize.py), forced DCT forms, and the small randomised parity tests with other seeds.
#   tools/soak_round.sh <tag> [processes] [count per process]
TAG=${1:-soak}; NP=${2:-8}; CNT=${3:-10}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
pids=()
for i in $(seq 1 $NP); do
  ( echo "== seed $((${SEED0:-600} + i)) count $CNT default kernels"; timeout 1500 python3 tools/soak_fullsize.py $((${SEED0:-600} + i)) $CNT 2>&1 | tail -$((CNT + 3)) ) > $OUT/full_$i.txt &
  pids+=($!)
done
for s in 0 2 6; do
  ( echo "== seed $((650 + s)) count 4 NLK_GROUP_SEP=$s"; NLK_GROUP_SEP=$s timeout 1500 python3 tools/soak_fullsize.py $((650 + s)) 4 2>&1 | tail -7 ) > $OUT/sep_$s.txt &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for seed in 701 702 703; do
  echo "== small configurations, 1000, seed $seed"
  NLK_RANDOM_SEED=$seed NLK_RANDOM_COUNT=1000 timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k randomised 2>&1 | tail -1
done > $OUT/small.txt
echo "== 300 random TV-L1 frames" >> $OUT/small.txt
NLK_RANDOM_SEED=77 NLK_RANDOM_COUNT=300 timeout 900 python3 -m pytest tests/test_tvl1.py -q -m gpu -k randomised 2>&1 | tail -1 >> $OUT/small.txt
cat $OUT/full_*.txt $OUT/sep_*.txt $OUT/small.txt > $OUT/soak_all.txt
grep -c "^ok" $OUT/soak_all.txt; grep -v "^ok" $OUT/soak_all.txt | tail -30
