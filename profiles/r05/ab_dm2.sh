#!/bin/bash
# same-box A/B of library builds on configs[1] with one order stream per replica (1024 days over 1024 replicas: day mode 2) and 512 days
#   bash profiles/r05/ab_dm2.sh <out.txt> <lib> <lib> ...
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
: > $OUT
for pass in 1 2; do
  for nd in 1024 512; do
    for lib in "$@"; do
      v=$(VDS_LIB=$PWD/$lib timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-neighbour-leg --distinct-days 2 --distinct-all-days $nd --no-hooked-leg --no-fallbacks-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['per_replica_days']['one_stream_per_replica']; print('%d days  %.3f ms/day  %.3e  %s' % (o['distinct_days'], o['ms_per_step'], o['value'], d['build']))")
      echo "pass $pass $lib $v" | tee -a $OUT
    done
  done
done
