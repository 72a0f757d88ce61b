#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c9.txt; : > $O
(timeout 900 python -m pytest tests/test_gpu_replica_days.py tests/test_gpu_parity.py tests/test_gpu_real_shape.py tests/test_gpu_full_size_properties.py tests/test_gpu_mutable_surface.py -x -q 2>&1 | tail -5) >> $O
(VDS_FUZZ_N=600 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3) >> $O
for D in 1 16 128; do timeout 300 python profiles/r04/probe_days2.py $D >> $O 2>&1; done
VDS_DENSE_LPR=16 timeout 300 python profiles/r04/probe_days2.py 1 >> $O 2>&1
for g in 1 2; do VDS_RUN_GROUPS=$g python bench.py --steps 30 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-distinct-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('groups=$g  %.3e  ms/day %.3f  tick us %.1f  one-chain us %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['one_chain_ms_per_tick']*1e3))" >> $O; done
VDS_DENSE=0 python bench.py --steps 30 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-distinct-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('k_tick_rows (same box) %.3e  ms/day %.3f  tick us %.1f  one-chain us %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['one_chain_ms_per_tick']*1e3))" >> $O
grep -v amdgpu.ids $O
