#!/bin/bash
# k_dfs_walk: up to three dry orders of one bucket share a scan (WK_G = 3; g2 = pairs as before), on top of v1 + the serve-loop cuts;
# k_reset_staged (lists built in LDS, written as runs) against k_reset_fast (VDS_RESET_UNSTAGED=1)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c23.txt; : > $O
B=$PWD/build
L=$PWD/vehicles_dispatch_simulator_amd/libvds.so
timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_dfs_shapes.py tests/test_gpu_real_shape.py tests/test_gpu_fuzz.py tests/test_gpu_run_groups.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3 >> $O
timeout 600 python profiles/full_check.py cfg4 256 2>&1 | tail -1 >> $O
timeout 600 python profiles/full_check.py cfg4 128 8 interleaved 2>&1 | tail -1 >> $O
timeout 600 python profiles/full_check.py cfg2 256 2>&1 | tail -1 >> $O
python profiles/ab.py $B/libvds_head.so $B/libvds_v1.so $B/libvds_g2.so $L --workload cfg4 --days 80 --rounds 3 >> $O 2>&1
python profiles/ab.py $B/libvds_head.so $L@VDS_RESET_UNSTAGED=1 $L --workload cfg2 --days 400 --rounds 3 >> $O 2>&1
echo "== sections (instrumented build of libvds.so's sources)" >> $O
VDS_LIB=$B/libvds_prof.so timeout 600 python profiles/sections_dfs.py 1024 >> $O 2>&1
grep -v amdgpu.ids $O | tail -40
