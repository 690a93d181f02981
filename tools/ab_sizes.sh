cd $GRAFT_REPO_ROOT
cp bwd-nlkalman_amd/libnlk_hip.so /tmp/cur.so
for L in bwd-nlkalman_amd/ab/*.so; do
  cp $L bwd-nlkalman_amd/libnlk_hip.so
  for sz in "640 480 3 20" "1280 720 3 20" "1920 1080 3 40" "3840 2160 3 20" "1280 720 1 20"; do
    echo "== $(basename $L) $sz"
    timeout 300 python3 tools/mode_times.py $sz 5 2>/dev/null | python3 -c "import sys,re
for l in sys.stdin:
    m=re.match(r'(.*?)\s+layout', l); g=dict(re.findall(r'(match|group|total) ([0-9.]+)', l))
    print('   ', m.group(1).strip().ljust(14), 'match', g['match'], 'group', g['group'], 'total', g['total'])"
  done
done
cp /tmp/cur.so bwd-nlkalman_amd/libnlk_hip.so
