#!/bin/bash
# k_dfs_walk: the steals each dry order saw through the visit sets as a bit matrix (Static.vis_bits) instead of per-cluster bitmaps
# built in LDS; eight visit rows in flight in the evaluation pass (a).  prev = the build of the previous commit.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c24.txt; : > $O
B=$PWD/build
L=$PWD/vehicles_dispatch_simulator_amd/libvds.so
timeout 1500 python -m pytest tests/test_gpu_dfs_shapes.py tests/test_gpu_real_shape.py tests/test_gpu_fuzz.py tests/test_gpu_run_groups.py tests/test_gpu_replica_days.py -x -q 2>&1 | tail -3 >> $O
timeout 600 python profiles/full_check.py cfg4 256 2>&1 | tail -1 >> $O
timeout 600 python profiles/full_check.py cfg4 128 8 interleaved 2>&1 | tail -1 >> $O
python profiles/ab.py $B/libvds_head.so $B/libvds_prev.so $L --workload cfg4 --days 80 --rounds 3 >> $O 2>&1
echo "== sections (instrumented build of libvds.so's sources)" >> $O
VDS_LIB=$B/libvds_prof.so timeout 600 python profiles/sections_dfs.py 1024 >> $O 2>&1
grep -v amdgpu.ids $O | tail -40
