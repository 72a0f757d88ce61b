#!/bin/bash
# bench lines of the final build of round 4 (after profiles/traffic.json / limiter.json carry its PMC figures)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
timeout 600 python bench.py --workload cfg5 --replicas 128 --steps 20 --warmup 3 --no-cpu-baseline --check > gpurun_out/r04_bench_cfg5.json 2>> gpurun_out/r04_bench.err
timeout 600 python bench.py --workload cfg4 --steps 10 --no-cpu-baseline --check --distinct-days 0 --no-distinct-all > gpurun_out/r04_bench_cfg4.json 2>> gpurun_out/r04_bench.err
timeout 600 python bench.py --workload cfg4 --replicas 2048 --steps 5 --no-cpu-baseline --distinct-days 0 --no-distinct-all > gpurun_out/r04_bench_cfg4_r2048.json 2>> gpurun_out/r04_bench.err
timeout 600 python profiles/r04/fallbacks.py 2>&1 | grep -v amdgpu > gpurun_out/r04_fallbacks.txt
for f in r04_bench r04_bench_cfg5 r04_bench_cfg4 r04_bench_cfg4_r2048; do echo "== $f"; cut -c1-400 gpurun_out/$f.json; done; cat gpurun_out/r04_fallbacks.txt; tail -3 gpurun_out/r04_bench.err
