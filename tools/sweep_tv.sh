cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 timeout 200 python3 bench.py --no-cpu --workload S1 --steps 12 --warmup 2 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run X=0; run NLK_TV_LOOK=2; run NLK_TV_LOOK=6; run NLK_TV_LOOK=8; run NLK_TV_INLINE=200; run NLK_TV_INLINE=1500; run NLK_TV_MID=0; run NLK_TV_MID=2; run NLK_TV_WG_PIXELS=2048; run NLK_TV_WG_PIXELS=16384; run X=0
