cd $GRAFT_REPO_ROOT
for gx in 2 3 4; do for gy in 1 2 3; do
  echo -n "GTX=$gx GTY=$gy: "; NLK_GTX=$gx NLK_GTY=$gy timeout 120 python3 tools/mode_times.py 2>&1 | grep "FLT1 spatial"
done; done
