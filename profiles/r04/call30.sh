#!/bin/bash
# host-only change (two run groups for large cities from 32 replicas on): run-group tests, the stress configuration's bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c30.txt; : > $O
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())" 2>/dev/null >> $O
timeout 900 python -m pytest tests/test_gpu_run_groups.py tests/test_gpu_stress_config.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3 >> $O
timeout 600 python bench.py --workload cfg5 --replicas 128 --steps 20 --warmup 3 --no-cpu-baseline --check > gpurun_out/r04_bench_cfg5.json 2>> gpurun_out/r04_bench.err
timeout 600 python bench.py --workload cfg5 --replicas 64 --steps 20 --warmup 3 --no-cpu-baseline --check > gpurun_out/r04_bench_cfg5_r64.json 2>> gpurun_out/r04_bench.err
cut -c1-300 gpurun_out/r04_bench_cfg5.json >> $O; cut -c1-300 gpurun_out/r04_bench_cfg5_r64.json >> $O
cat $O
