#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c6.txt; : > $O
(timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_simulation_shell.py tests/test_gpu_debug_builds.py -x -q 2>&1 | tail -5) >> $O
for rep in 1 2; do
for lib in "" build/libvds_w5.so build/libvds_w4.so; do
  for g in 1 2; do
    VDS_LIB=${lib:+$PWD/$lib} VDS_RUN_GROUPS=$g python bench.py --steps 30 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-distinct-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=%s groups=$g  %.3e  ms/day %.3f  tick us %.1f  one-chain us %.1f' % ('$lib' or 'default(6 waves)', d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['one_chain_ms_per_tick']*1e3))" >> $O
  done
done
done
cat $O
