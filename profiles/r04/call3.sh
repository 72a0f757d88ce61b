#!/bin/bash
# gpurun call 3 of round 4: per-row order days on the dense tick (day mode 2), 8 lanes per replica by default
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_replica_days.py tests/test_gpu_parity.py tests/test_gpu_run_groups.py -x -q 2>&1 | tail -15) > gpurun_out/r04_c3_tests.txt
(VDS_FUZZ_N=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8) >> gpurun_out/r04_c3_tests.txt
timeout 900 python bench.py > gpurun_out/r04_bench3.json 2> gpurun_out/r04_bench3.err
cat gpurun_out/r04_c3_tests.txt; cut -c1-400 gpurun_out/r04_bench3.json; tail -3 gpurun_out/r04_bench3.err
