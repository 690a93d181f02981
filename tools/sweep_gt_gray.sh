cd $GRAFT_REPO_ROOT
for cfg in "0 0" "3 2" "2 2" "4 2" "3 3" "3 1" "4 1" "6 1" "6 2"; do
  set -- $cfg
  if [ $1 = 0 ]; then echo "default gray"; timeout 300 python3 tools/mode_times.py 1920 1080 1 20 2>/dev/null | awk '{print "   ", $1, $2, $9, $10}';
  else echo "GTX=$1 GTY=$2 gray"; NLK_GTX=$1 NLK_GTY=$2 timeout 300 python3 tools/mode_times.py 1920 1080 1 20 2>/dev/null | awk '{print "   ", $1, $2, $9, $10}'; fi
done
