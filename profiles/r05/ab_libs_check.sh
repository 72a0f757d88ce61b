#!/bin/bash
# as ab_libs.sh, with the parity check of replica 0 against the oracle in every run
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
: > $OUT
for pass in 1 2 3; do
  for lib in "$@"; do
    v=$(VDS_LIB=$PWD/$lib timeout 200 python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-fallbacks-leg --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/day  one chain %.3f ms  check %s  %s' % (d['ms_per_step'], d['roofline']['one_chain_day_kernel_ms'], d.get('parity_check_vs_oracle'), d['build']))")
    echo "pass $pass $lib $v" | tee -a $OUT
  done
done
