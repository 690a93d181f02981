#!/bin/bash
# The whole evidence run in ONE gpurun call, so that the bench lines carry `roofline.traffic` of exactly the
# binaries they timed: counter passes (C2, C3, F1) -> traffic table (written into the box's own profiles/) ->
# bench lines, strip model, A/B files.   tools/evidence_all.sh <tag>      then, back home:
#   tools/collect_profiles.sh <tag> r06   (copies the summaries; re-makes the table with the git head in it)
set -u
TAG=${1:-r06x}
RND=${2:-r06}
cd $GRAFT_REPO_ROOT
timeout 2400 bash tools/profile_pmc_only.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
timeout 300 bash tools/profile_pmc_f1.sh $TAG >> gpurun_out/profile_$TAG.log 2>&1
mkdir -p gpurun_out/$RND
cp gpurun_out/$TAG/pmc_*_summary.txt gpurun_out/$RND/
python3 tools/make_traffic.py $RND >> gpurun_out/profile_$TAG.log 2>&1
cp profiles/${RND}_traffic.json gpurun_out/$TAG/traffic_made_on_the_box.json
LINES_ONLY=1 timeout 2300 bash tools/profile_round.sh $TAG >> gpurun_out/profile_$TAG.log 2>&1
ls gpurun_out/$TAG | wc -l
