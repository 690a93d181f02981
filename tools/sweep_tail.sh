for t in 8 16 24 32; do for s in 3 6 10; do
  NLK_G8_TAIL=$t NLK_G8_SINGLE=$s python bench.py --steps 60 --no-cpu --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('tail $t single $s', d['ms_per_step'], d['kernels_ms']['group_ms'])"
done; done
