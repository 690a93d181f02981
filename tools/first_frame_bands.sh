set -u
python -m pytest tests/test_gpu_parity.py -x -q -k "mask_replay or banded or spatial or first" 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -x -q -k "spatial or reach2 or nan_ring" 2>&1 | tail -3
for nb in 1 2 4; do
  for i in 1 2; do
    NLK_BANDS=$nb python bench.py --steps 30 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('bands $nb first_frame_ms', d['first_frame_ms'], 'ms_per_step', d['ms_per_step'])"
  done
done
NLK_COMMIT_WAVE=1 python bench.py --steps 30 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('wave(4 bands) first_frame_ms', d['first_frame_ms'], 'ms_per_step', d['ms_per_step'])"
python tools/mode_times.py 2>/dev/null | tail -12 || true
