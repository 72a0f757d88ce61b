#!/bin/bash
# one gpurun call of round 4: graph-pool life cycle, byte calibration, lanes-per-replica A/B, bench with the new legs
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_run_groups.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -8) > gpurun_out/r04_pool_tests.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_run_groups.py -x -q -k life_cycle 2>&1 | tail -2 >> gpurun_out/r04_pool_tests.txt; done
for i in 1 2 3 4; do timeout 120 python profiles/r03_run_groups/crash_probe.py 3 2 3 12 2>&1 | tail -1 >> gpurun_out/r04_pool_tests.txt; done
bash profiles/ubench/bytes_calib.sh > gpurun_out/bytes_calib.log 2>&1
timeout 600 python bench.py --no-distinct-all > gpurun_out/r04_bench2.json 2> gpurun_out/r04_bench2.err
bash profiles/r04/ab_lpr.sh 20 > gpurun_out/r04_ab_lpr.txt 2>&1
cat gpurun_out/r04_pool_tests.txt; cat gpurun_out/r04_ab_lpr.txt
