#!/bin/bash
# End-of-round evidence run (gpurun): kernel-trace statistics of the C2 and F1 bench commands and
# the FETCH_SIZE / WRITE_SIZE passes of F1 (separate --pmc runs, kernel-trace only).
#   tools/profile_round.sh <tag>
set -u
TAG=${1:-r01_h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 $GRAFT_REPO_ROOT/bench.py --workload F1 > $OUT/bench_f1.json 2> $OUT/bench_f1.err
python3 $GRAFT_REPO_ROOT/bench.py --workload S1 --steps 10 --warmup 2 > $OUT/bench_s1.json 2> $OUT/bench_s1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $OUT/bench_c2_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f1 -o f1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload F1 --no-cpu > $OUT/bench_f1_under_rocprof.json 2>/dev/null
for SET in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$SET -o f1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload F1 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
find $OUT -name "*.csv" | head -20
# keep the merged output small: the raw traces are large
find $OUT -name "*kernel_trace.csv" -size +8M -delete
