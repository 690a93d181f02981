#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run from gpurun_out/<src> (scratch) into profiles/<tag>_* (tracked)
#   tools/collect_profiles.sh <src> <tag>      e.g.  tools/collect_profiles.sh r03a r03
set -eu
SRC=gpurun_out/$1; TAG=$2
cd "$(dirname "$0")/.."
for f in $SRC/bench_*.json; do
  b=$(basename $f .json)
  [ -s $f ] && cp $f profiles/${TAG}_$b.json
done
for W in C2 C3 C5 C1L; do
  [ -f $SRC/stats_$W/s_kernel_stats.csv ] && cp $SRC/stats_$W/s_kernel_stats.csv profiles/${TAG}_$(echo $W | tr A-Z a-z)_kernel_stats.csv
  [ -f $SRC/pmc_${W}_summary.txt ] && cp $SRC/pmc_${W}_summary.txt profiles/${TAG}_pmc_${W}_summary.txt
done
[ -f $SRC/pmc_F1_summary.txt ] && cp $SRC/pmc_F1_summary.txt profiles/${TAG}_pmc_F1_summary.txt
[ -f $SRC/mode_times_1080p.txt ] && cp $SRC/mode_times_1080p.txt profiles/${TAG}_mode_times_1080p.txt
[ -f $SRC/api_wall.txt ] && cp $SRC/api_wall.txt profiles/${TAG}_api_wall.txt
[ -f $SRC/startup_times.txt ] && cp $SRC/startup_times.txt profiles/${TAG}_startup_times.txt
[ -f $SRC/power_probe.txt ] && cp $SRC/power_probe.txt profiles/${TAG}_power_probe.txt
for f in strip_model_phases ab_no_chase first_frame_bands mode_times_1080p_gray ab_group_sep mode_times_1080p_by_sep mode_times_1080p_gray_by_sep ab_match_order; do
  [ -f $SRC/$f.txt ] && cp $SRC/$f.txt profiles/${TAG}_$f.txt
done
# the traffic table reads gpurun_out/<tag>/pmc_*_summary.txt
mkdir -p gpurun_out/$TAG
cp $SRC/pmc_*_summary.txt gpurun_out/$TAG/
python3 tools/make_traffic.py $TAG
ls profiles | grep "^${TAG}_" | tr '\n' ' '
