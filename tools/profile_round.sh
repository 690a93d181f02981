#!/bin/bash
# End-of-round evidence run (gpurun): bench lines, rocprofv3 kernel statistics and PMC passes
# (separate --pmc runs, kernel-trace only, program directly after `--`) for the C2, C3 and F1 workloads.
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/ ; then tools/make_traffic.py <tag>
# Every command runs under `timeout` (a profiler that does not come back must not eat the GPU budget).
set -u
TAG=${1:-r06}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout 300"
for W in C2 C3 C1 C1L; do $T python3 $ROOT/bench.py --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err; done
$T python3 $ROOT/bench.py --workload C5 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
$T python3 $ROOT/bench.py --workload F1 > $OUT/bench_F1.json 2> $OUT/bench_F1.err
$T python3 $ROOT/bench.py --workload S1 --steps 10 --warmup 2 > $OUT/bench_S1.json 2> $OUT/bench_S1.err
NLK_MATCH_ORDER=block $T python3 $ROOT/bench.py > $OUT/bench_C2_match_order_block.json 2>/dev/null
NLK_DETERMINISTIC=1 $T python3 $ROOT/bench.py --no-cpu > $OUT/bench_C2_deterministic.json 2>/dev/null
NLK_DETERMINISTIC=1 $T python3 $ROOT/bench.py --no-cpu --workload C3 > $OUT/bench_C3_deterministic.json 2>/dev/null
# the N > 1 code path of bench.py on the one GPU of this box (every rank on device 0 over gloo): plumbing, NOT a measurement
NLK_BENCH_ONE_GPU=1 timeout 600 python3 $ROOT/bench.py --gpus 2 --steps 5 --warmup 2 --phase-times > $OUT/bench_C2_2ranks_onegpu.json 2> $OUT/bench_C2_2ranks_onegpu.err
NLK_BENCH_ONE_GPU=1 timeout 900 python3 $ROOT/bench.py --gpus 8 --steps 5 --warmup 2 --phase-times > $OUT/bench_C2_8ranks_onegpu.json 2> $OUT/bench_C2_8ranks_onegpu.err
# the N > 1 step at N = 1 (--force-strips): enqueued from C (plain launches / replayed HIP graph) and from Python,
# with the per-phase device times: what the three-phase machinery costs over the whole-frame call
for W in C2 C3; do
  $T python3 $ROOT/bench.py --no-cpu --workload $W --force-strips --phase-times > $OUT/bench_${W}_force_strips_c.json 2>/dev/null
  $T python3 $ROOT/bench.py --no-cpu --workload $W --force-strips --phase-times --strip-graph > $OUT/bench_${W}_force_strips_c_graph.json 2>/dev/null
  $T python3 $ROOT/bench.py --no-cpu --workload $W --force-strips --phase-times --strip-driver py > $OUT/bench_${W}_force_strips_py.json 2>/dev/null
done
$T python3 $ROOT/tools/startup_times.py > $OUT/startup_times.txt 2>&1
$T python3 $ROOT/tools/power_probe.py --steps 3000 --warmup 50 --no-cpu > $OUT/power_probe.txt 2>&1
$T python3 $ROOT/tools/mode_times.py > $OUT/mode_times_1080p.txt 2>&1
$T python3 $ROOT/tools/mode_times.py 1920 1080 1 20 > $OUT/mode_times_1080p_gray.txt 2>&1
(cd $ROOT && bash tools/ab_sep.sh C2 2) > $OUT/ab_group_sep.txt 2>&1
(cd $ROOT && bash tools/ab_env.sh NLK_MATCH_ORDER=block C2 2) > $OUT/ab_match_order.txt 2>&1
for s in 0 2 6; do echo "NLK_GROUP_SEP=$s"; NLK_GROUP_SEP=$s $T python3 $ROOT/tools/mode_times.py 1920 1080 1 20 2>/dev/null | grep layout; done > $OUT/mode_times_1080p_gray_by_sep.txt 2>&1
for s in 0 2 6; do echo "NLK_GROUP_SEP=$s"; NLK_GROUP_SEP=$s $T python3 $ROOT/tools/mode_times.py 2>/dev/null | grep layout; done > $OUT/mode_times_1080p_by_sep.txt 2>&1
NLK_HOST_TRACE=1 $T python3 $ROOT/tools/api_wall.py > $OUT/api_wall.txt 2>&1
# the one-rank-of-N model (exchanges skipped) and where its step goes
for W in C2 C3; do
  $T python3 $ROOT/bench.py --no-cpu --workload $W --strip-model > $OUT/bench_${W}_strip_model.json 2>/dev/null
done
$T python3 $ROOT/tools/strip_model_phases.py 1 2 4 8 > $OUT/strip_model_phases.txt 2>&1
(cd $ROOT && bash tools/ab_env.sh NLK_NO_CHASE=1 C2 3) > $OUT/ab_no_chase.txt 2>&1
(cd $ROOT && bash tools/first_frame_bands.sh) > $OUT/first_frame_bands.txt 2>&1
# LINES_ONLY=1: stop here (the counters were collected before, tools/profile_pmc_only.sh, so that these lines
# could carry roofline.traffic of the same sources)
[ -n "${LINES_ONLY:-}" ] && exit 0
for W in C2 C3; do
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
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  $T rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_F1/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --workload F1 > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_F1 > $OUT/pmc_F1_summary.txt 2>&1
find $OUT -name "*_kernel_stats.csv" | head
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
