#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c11.txt; : > $O
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replica_days.py -x -q 2>&1 | tail -3) >> $O
python profiles/ab.py $PWD/build/libvds_prev.so $PWD/vehicles_dispatch_simulator_amd/libvds.so --days 400 --rounds 3 >> $O 2>&1
python profiles/ab.py $PWD/build/libvds_prev.so $PWD/vehicles_dispatch_simulator_amd/libvds.so --days 200 --rounds 2 --distinct 16 >> $O 2>&1
VDS_LIB=$PWD/build/libvds_prof.so VDS_RUN_GRAPH=0 timeout 300 python profiles/r04/sections_dense.py 1024 1 >> $O 2>&1
VDS_LIB=$PWD/build/libvds_prof.so timeout 300 python profiles/r04/inflight.py >> $O 2>&1
grep -v amdgpu.ids $O
