# Tile shapes of k_tv_block on the levels that do not fill the chip (tvl1_host.h: NLK_TV_S1 / S2 / S3 = shape for
# levels of >= 100 / >= 20 / fewer 64 x 16 tiles), one 1080p flow at fscale 1:  tools/sweep_tv_small.sh
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout 200 python3 tools/tvl1_time.py 1920 1080 1 2>/dev/null | head -1; }
run X=0
if [ "${1:-}" = combos ]; then
  run NLK_TV_S2=7 NLK_TV_S3=7; run NLK_TV_S2=8 NLK_TV_S3=7; run NLK_TV_S2=4 NLK_TV_S3=7; run NLK_TV_S1=4 NLK_TV_S2=8 NLK_TV_S3=7
  run NLK_TV_S2=8 NLK_TV_S3=7 NLK_TV_LOOK=3; run NLK_TV_S2=8 NLK_TV_S3=7 NLK_TV_LOOK=6; run NLK_TV_S2=8 NLK_TV_S3=7 NLK_TV_WG_PIXELS=1000
else
  for v in S1 S2 S3; do for s in 4 5 6 7 8 9; do run NLK_TV_$v=$s; done; done
fi
run X=0
