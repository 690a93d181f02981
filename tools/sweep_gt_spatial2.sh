cd $GRAFT_REPO_ROOT
for cfg in "0 0" "2 2" "3 2" "2 3"; do
  set -- $cfg
  for sz in "1920 1080 1 20" "3840 2160 3 20"; do
  if [ $1 = 0 ]; then echo -n "default $sz: "; timeout 300 python3 tools/mode_times.py $sz 3 2>/dev/null | grep spatial | awk '{print $9, $10, $13, $14, $15, $16}';
  else echo -n "GTX=$1 GTY=$2 $sz: "; NLK_GTX=$1 NLK_GTY=$2 timeout 300 python3 tools/mode_times.py $sz 3 2>/dev/null | grep spatial | awk '{print $9, $10, $13, $14, $15, $16}'; fi
  done
done
