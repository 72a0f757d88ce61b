#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c12.txt; : > $O
VDS_LIB=$PWD/build/libvds_prof.so timeout 300 python profiles/r04/inflight.py >> $O 2>&1
for g in 1 2 3 4; do VDS_RUN_GROUPS=$g python bench.py --steps 40 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-distinct-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('groups=$g  %.3e  ms/day %.3f  tick us %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3))" >> $O; done
for R in 128 256 512; do timeout 600 python bench.py --workload cfg5 --replicas $R --steps 5 --warmup 1 --no-cpu-baseline --check 2>>$O | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg5 R=$R  %.3e  ms/day %.3f  tick us %.1f one-chain %.1f kernel %s check %s slow %d' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['one_chain_ms_per_tick']*1e3, r['kernel'], d.get('parity_check_vs_oracle'), d['slow_path_buckets_last_day']))" >> $O; done
grep -v amdgpu.ids $O
