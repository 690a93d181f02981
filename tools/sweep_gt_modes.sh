#!/bin/bash
# group-kernel tile shapes by mode (tools/mode_times.py): tools/sweep_gt_modes.sh
cd $GRAFT_REPO_ROOT
for cfg in "3 2" "3 1" "2 2" "4 2" "3 3" "4 1" "2 1"; do
  set -- $cfg
  echo "GTX=$1 GTY=$2"
  NLK_GTX=$1 NLK_GTY=$2 timeout 300 python3 tools/mode_times.py 2>/dev/null | awk '{print "   ", $1, $2, $9, $10}'
done
