#!/bin/bash
# A/B of library variants (bwd-nlkalman_amd/ab/*.so) on the four frame calls:  tools/ab_modes.sh [mode_times args]
cd $GRAFT_REPO_ROOT
cp bwd-nlkalman_amd/libnlk_hip.so /tmp/libnlk_hip_current.so
for rep in 1 2; do
for L in bwd-nlkalman_amd/ab/*.so; do
  cp $L bwd-nlkalman_amd/libnlk_hip.so
  echo "== $(basename $L) (run $rep)"
  python3 tools/mode_times.py "$@" 2>/dev/null | grep "layout"
done
done
cp /tmp/libnlk_hip_current.so bwd-nlkalman_amd/libnlk_hip.so
