#!/bin/bash
# first frame at 1080p (deno0 = NULL): default (replay inside the group launch, one band) against the separate replay
# kernels in 1 / 2 / 4 bands; bench.py's first_frame_ms and tools/mode_times.py's per-phase times (one band, events)
for e in "" "NLK_NO_CHASE=1" "NLK_NO_CHASE=1 NLK_BANDS=1" "NLK_NO_CHASE=1 NLK_BANDS=2" "NLK_BANDS=4"; do
  for i in 1 2; do
    env $e timeout 200 python bench.py --steps 30 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('${e:-default}'.ljust(28), 'first_frame_ms', d['first_frame_ms'], 'ms_per_step', d['ms_per_step'])"
  done
done
timeout 200 python tools/mode_times.py 2>/dev/null | tail -6
