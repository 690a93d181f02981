#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh NLK_NO_CHASE=1 [workload] [rounds]
# prints ms_per_step / group_ms / commit_ms of the default and of the variant, interleaved
V=$1; W=${2:-C2}; N=${3:-3}
for i in $(seq $N); do
  for e in "" "$V"; do
    env $e python bench.py --workload $W --steps 50 --no-cpu --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernels_ms']
print('${e:-default}'.ljust(24), 'ms_per_step', d['ms_per_step'], 'group', k['group_ms'], 'commit', k['commit_ms'], 'match', k['match_ms'], 'total', k['total_ms'])"
  done
done
