#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c7.txt; : > $O
(timeout 900 python -m pytest tests/test_gpu_replica_days.py tests/test_gpu_run_groups.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5) >> $O
(VDS_FUZZ_N=400 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3) >> $O
for D in 16 128; do timeout 300 python profiles/r04/probe_days2.py $D >> $O 2>&1; VDS_DENSE_TAB256=0 timeout 300 python profiles/r04/probe_days2.py $D >> $O 2>&1; done
timeout 300 python profiles/r04/probe_days2.py 1 >> $O 2>&1
VDS_LIB=$PWD/build/libvds_prof.so VDS_RUN_GRAPH=0 timeout 300 python profiles/r04/sections_dense.py >> $O 2>&1
grep -v amdgpu.ids $O
