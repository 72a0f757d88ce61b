#!/bin/bash
# same-box A/B of library builds on the headline (configs[1], 1024 replicas): bash profiles/r05/ab_libs.sh <out.txt> <lib> <lib> ...
# (each arm 300 timed days, the arms interleaved, three passes)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
: > $OUT
for pass in 1 2 3; do
  for lib in "$@"; do
    v=$(VDS_LIB=$PWD/$lib timeout 200 python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-fallbacks-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/day  one chain %.3f ms  %s' % (d['ms_per_step'], d['roofline']['one_chain_day_kernel_ms'], d['build']))")
    echo "pass $pass $lib $v" | tee -a $OUT
  done
done
