cd $GRAFT_REPO_ROOT
for E in "NLK_GROUP_SEP=0" "NLK_X=1"; do
  echo "== $E"; env $E python3 tools/mode_times.py 2>/dev/null
  echo "== gray $E"; env $E python3 tools/mode_times.py 1920 1080 1 20 2>/dev/null
done
