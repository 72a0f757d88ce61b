#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c16.txt; : > $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_run_groups.py tests/test_gpu_real_shape.py tests/test_gpu_full_size_properties.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3) >> $O
(VDS_FUZZ_N=400 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2) >> $O
python profiles/ab.py $PWD/build/libvds_prev.so $PWD/build/libvds_td.so $PWD/build/libvds_r32.so --days 400 --rounds 3 >> $O 2>&1
VDS_RUN_GROUPS=1 python profiles/ab.py $PWD/build/libvds_prev.so $PWD/build/libvds_td.so $PWD/build/libvds_r32.so --days 300 --rounds 2 >> $O 2>&1
grep -v amdgpu.ids $O
