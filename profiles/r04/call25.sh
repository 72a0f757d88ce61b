#!/bin/bash
# k_dfs_walk: compaction pass software-pipelined (the next batch's loads before this batch's stores).  evalb = the build before.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c25.txt; : > $O
B=$PWD/build
L=$PWD/vehicles_dispatch_simulator_amd/libvds.so
timeout 1500 python -m pytest tests/test_gpu_dfs_shapes.py tests/test_gpu_real_shape.py tests/test_gpu_fuzz.py tests/test_gpu_run_groups.py -x -q 2>&1 | tail -3 >> $O
timeout 600 python profiles/full_check.py cfg4 256 2>&1 | tail -1 >> $O
python profiles/ab.py $B/libvds_head.so $B/libvds_evalb.so $L --workload cfg4 --days 80 --rounds 3 >> $O 2>&1
echo "== sections (instrumented build of libvds.so's sources)" >> $O
VDS_LIB=$B/libvds_prof.so timeout 600 python profiles/sections_dfs.py 1024 >> $O 2>&1
grep -v amdgpu.ids $O | tail -40
