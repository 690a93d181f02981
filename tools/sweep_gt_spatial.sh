cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 2" "2 2" "2 3" "1 3" "1 4" "2 4" "3 2" "2 1" "1 1"; do
  set -- $cfg
  if [ $1 = 0 ]; then echo -n "default: "; timeout 300 python3 tools/mode_times.py 2>/dev/null | grep spatial | awk '{print $9, $10, $13, $14, $15, $16}';
  else echo -n "GTX=$1 GTY=$2: "; NLK_GTX=$1 NLK_GTY=$2 timeout 300 python3 tools/mode_times.py 2>/dev/null | grep spatial | awk '{print $9, $10, $13, $14, $15, $16}'; fi
done
