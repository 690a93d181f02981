cd $GRAFT_REPO_ROOT
for cfg in "3 2" "3 1" "2 2" "4 2" "3 3" "2 3" "4 1" "6 1"; do
  set -- $cfg
  echo -n "GTX=$1 GTY=$2: "
  NLK_GTX=$1 NLK_GTY=$2 timeout 300 python3 bench.py --no-cpu --no-extras --steps 30 --workload C2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms']['group_ms'])"
done
