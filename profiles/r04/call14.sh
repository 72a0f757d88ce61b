#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c14.txt; : > $O
(timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_stress_config.py tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -5) >> $O
VDS_LIB=$PWD/build/libvds_prof.so timeout 300 python profiles/r04/inflight.py >> $O 2>&1
for W in "cfg5 128" "cfg5 256" "cfg2 1024"; do set -- $W
  timeout 600 python bench.py --workload $1 --replicas $2 --steps 10 --warmup 3 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-distinct-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 R=$2  %.3e  ms/day %.3f  tick us %.1f one-chain %.1f kernel %s slow %d' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['one_chain_ms_per_tick']*1e3, r['kernel'], d['slow_path_buckets_last_day']))" >> $O
done
grep -v amdgpu.ids $O
