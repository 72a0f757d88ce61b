#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c17.txt; : > $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replica_days.py tests/test_gpu_mutable_surface.py tests/test_gpu_simulation_shell.py tests/test_gpu_edge_cases.py tests/test_gpu_real_shape.py -x -q 2>&1 | tail -3) >> $O
(VDS_FUZZ_N=600 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2) >> $O
python profiles/ab.py $PWD/build/libvds_prev.so $PWD/build/libvds_nif8.so $PWD/build/libvds_arr32.so --days 400 --rounds 3 >> $O 2>&1
python profiles/ab.py $PWD/build/libvds_nif8.so $PWD/build/libvds_arr32.so --days 200 --rounds 2 --distinct 16 >> $O 2>&1
python profiles/ab.py $PWD/build/libvds_nif8.so $PWD/build/libvds_arr32.so --days 100 --rounds 2 --distinct 128 >> $O 2>&1
grep -v amdgpu.ids $O
