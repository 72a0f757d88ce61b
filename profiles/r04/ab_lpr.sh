#!/bin/bash
# A/B of the dense tick's lanes-per-replica settings against the wide row-mapped kernel, one box, same process order twice
# (bench.py headline only).   bash profiles/r04/ab_lpr.sh [steps]
N=${1:-40}
for rep in 1 2; do
for cfg in "VDS_DENSE_LPR=16" "VDS_DENSE_LPR=8" "VDS_DENSE_LPR=4" "VDS_DENSE=0"; do
  for g in 1 2; do
    env $cfg VDS_RUN_GROUPS=$g python bench.py --steps $N --no-cpu-baseline --no-neighbour-leg --distinct-days 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg groups=$g  %.3e  ms/day %.3f  tick us %.1f  kernel %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['kernel']))"
  done
done
done
