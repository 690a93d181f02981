#!/bin/bash
# the group kernel's four DCT variants (NLK_GROUP_SEP = 0..3) on one box:  tools/ab_sep.sh [workload] [rounds]
W=${1:-C2}; N=${2:-2}
for i in $(seq $N); do
  for e in 0 2 6; do
    NLK_GROUP_SEP=$e python bench.py --workload $W --steps 50 --no-cpu --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernels_ms']
print('NLK_GROUP_SEP=$e', 'ms_per_step', d['ms_per_step'], 'group', k['group_ms'], 'match', k['match_ms'], 'total', k['total_ms'])"
  done
done
