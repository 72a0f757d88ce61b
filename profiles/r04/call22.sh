#!/bin/bash
# walk variants on one box + exactness.  v1: resolve with the ring atomics beside the vehicle-id gathers + batched compaction of
# 65..128-entry lists; v3 (libvds.so) = v1 + serve loop (lazy q, position in the record, ballot next_dry, compiler-only DS ordering)
# + cost-row touch in dfs_scan; v3fence / v3nopf / v3prio / v3w5: one switch each (waits back, no touch, s_setprio 3, 320 threads
# compiled for 5 wavefronts per SIMD)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c22.txt; : > $O
B=$PWD/build
for l in $PWD/vehicles_dispatch_simulator_amd/libvds.so $B/libvds_v3w5.so; do
  echo "== $l" >> $O
  VDS_LIB=$l timeout 900 python -m pytest tests/test_gpu_dfs_shapes.py tests/test_gpu_real_shape.py -x -q 2>&1 | tail -2 >> $O
  VDS_LIB=$l timeout 600 python profiles/full_check.py cfg4 128 2>&1 | tail -2 >> $O
done
python profiles/ab.py $B/libvds_head.so $B/libvds_v1.so $PWD/vehicles_dispatch_simulator_amd/libvds.so $B/libvds_v3fence.so $B/libvds_v3nopf.so $B/libvds_v3prio.so $B/libvds_v3w5.so --workload cfg4 --days 80 --rounds 3 >> $O 2>&1
echo "== one group" >> $O
VDS_RUN_GROUPS=1 python profiles/ab.py $B/libvds_head.so $PWD/vehicles_dispatch_simulator_amd/libvds.so $B/libvds_v3w5.so --workload cfg4 --days 60 --rounds 2 >> $O 2>&1
echo "== sections (instrumented build of libvds.so's sources)" >> $O
VDS_LIB=$B/libvds_prof.so timeout 600 python profiles/sections_dfs.py 1024 >> $O 2>&1
grep -v amdgpu.ids $O | tail -40
