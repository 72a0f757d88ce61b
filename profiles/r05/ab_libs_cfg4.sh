#!/bin/bash
# same-box A/B of library builds on configs[3] (neighbour search, 1024 replicas): bash profiles/r05/ab_libs_cfg4.sh <out.txt> <lib> ...
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
: > $OUT
for pass in 1 2; do
  for lib in "$@"; do
    v=$(VDS_LIB=$PWD/$lib timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 2 --no-cpu-baseline --distinct-days 0 --no-hooked-leg --no-fallbacks-leg --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/day  %.3e  check %s' % (d['ms_per_step'], d['value'], d.get('parity_check_vs_oracle')))")
    echo "pass $pass $lib $v" | tee -a $OUT
  done
done
