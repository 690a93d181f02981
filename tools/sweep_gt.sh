# group tile shapes (targets per workgroup) against the four kinds of frame call at 1080p: tools/sweep_gt.sh (gpurun)
cd $GRAFT_REPO_ROOT
for gx in 2 3 4; do for gy in 1 2 3; do
  echo "GTX=$gx GTY=$gy"; NLK_GTX=$gx NLK_GTY=$gy timeout 120 python3 tools/mode_times.py 2>&1 | sed -E 's/layout.*group ([0-9.]+).*wall ([0-9.]+)/group \1 wall \2/'
done; done
