#!/bin/bash
# rocprofv3 kernel statistics of the runs the bench line does not cover: the four frame calls at 1080p
# (tools/mode_times.py: first frame, temporal, second iteration, smoother) and one rank of 8 stepped alone
# (tools/strip_model_phases.py 8).   tools/profile_extra_stats.sh <tag>
set -u
TAG=${1:-r05x}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_modes -o s -- python3 $ROOT/tools/mode_times.py > $OUT/mode_times_under_rocprof.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_strip8 -o s -- python3 $ROOT/tools/strip_model_phases.py 8 > $OUT/strip8_under_rocprof.txt 2>/dev/null
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
ls $OUT
