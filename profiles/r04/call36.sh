#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c36.txt; : > $O
timeout 300 python examples/batched_dispatch_loop.py 128 2>&1 | grep -v amdgpu | tail -4 >> $O
timeout 300 python examples/demo_simulation.py 2>&1 | grep -v amdgpu | tail -4 >> $O
VDS_FUZZ_N=6000 VDS_FUZZ_MEDIUM_N=1500 VDS_FUZZ_DAYS_N=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2 >> $O
cat $O
