#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c35.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_debug_builds.py tests/test_gpu_two_ranks.py tests/test_gpu_stress_config.py tests/test_gpu_full_size_properties.py tests/test_gpu_neighbors.py tests/test_gpu_primitives.py tests/test_gpu_real_shape.py tests/test_gpu_run_groups.py tests/test_gpu_dfs_shapes.py -q 2>&1 | grep -v amdgpu | tail -60 >> $O
cat $O
