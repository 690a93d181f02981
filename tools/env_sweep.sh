#!/bin/bash
# Timing of one environment switch's values on ONE box:  tools/env_sweep.sh VAR "v1 v2 ..." [bench args...]
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
for v in $VALS; do
  echo -n "$VAR=$v (run $rep): "
  env $VAR=$v timeout 300 python3 bench.py --no-cpu --steps 30 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v, 4) for k, v in d['kernels_ms'].items()})"
done
done
