#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
timeout 600 python bench.py --workload cfg5 --replicas 128 --steps 20 --warmup 3 --no-cpu-baseline --check > gpurun_out/r04_bench_cfg5.json 2>> gpurun_out/r04_bench.err
timeout 600 python bench.py --workload cfg4 --steps 10 --no-cpu-baseline --check --distinct-days 0 --no-distinct-all > gpurun_out/r04_bench_cfg4.json 2>> gpurun_out/r04_bench.err
timeout 600 python profiles/r04/fallbacks.py 2>&1 | grep -v amdgpu > gpurun_out/r04_fallbacks.txt
cut -c1-300 gpurun_out/r04_bench.json; cat gpurun_out/r04_fallbacks.txt; tail -3 gpurun_out/r04_bench.err
