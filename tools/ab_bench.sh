#!/bin/bash
# A/B timing of kernel variants on ONE box (boxes differ by a few per cent): every library saved under
# bwd-nlkalman_amd/ab/*.so (copies of libnlk_hip.so built from different sources; not tracked) is put in
# place in turn and bench.py run with the given arguments.   tools/ab_bench.sh [bench args...]
cd $GRAFT_REPO_ROOT
cp bwd-nlkalman_amd/libnlk_hip.so /tmp/libnlk_hip_current.so
for rep in 1 2; do
for L in bwd-nlkalman_amd/ab/*.so; do
  cp $L bwd-nlkalman_amd/libnlk_hip.so
  echo -n "$(basename $L) (run $rep): "
  python3 bench.py --no-cpu --steps 30 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v, 4) for k, v in d['kernels_ms'].items()})"
done
done
cp /tmp/libnlk_hip_current.so bwd-nlkalman_amd/libnlk_hip.so
