#!/bin/bash
# rocprofv3 kernel stats of the hybrid neighbour-search tick, one chain (VDS_RUN_GROUPS=1): deferred acceptance vs the serial walk, one box
#   bash profiles/r05/walk_stats.sh <tag>
export TMPDIR=/tmp
TAG=${1:-walk_stats}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
export VDS_RUN_GROUPS=1
B="python bench.py --workload cfg4 --no-cpu-baseline --steps 2 --warmup 1"
for mode in da serial; do
  if [ $mode = serial ]; then export VDS_WALK_DA=0; else unset VDS_WALK_DA; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -- $B > $O/$mode.log 2>&1
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; head -6 $f | cut -c1-200
done
