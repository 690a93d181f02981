#!/bin/bash
# tools/mode_times.py for every library under bwd-nlkalman_amd/ab/*.so, with the given environment:  tools/ab_modes_env.sh "NLK_GROUP_SEP=2" [mode_times args]
cd $GRAFT_REPO_ROOT
ENVV=$1; shift
cp bwd-nlkalman_amd/libnlk_hip.so /tmp/libnlk_hip_current.so
for L in bwd-nlkalman_amd/ab/*.so; do
  cp $L bwd-nlkalman_amd/libnlk_hip.so
  echo "== $(basename $L) $ENVV"
  env $ENVV python3 tools/mode_times.py "$@" 2>/dev/null
done
cp /tmp/libnlk_hip_current.so bwd-nlkalman_amd/libnlk_hip.so
