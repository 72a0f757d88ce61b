#!/bin/bash
# vds_reset_random (start nodes drawn on the device): exactness against CPython's random, timing against host generation + upload
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c32.txt; : > $O
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())" 2>/dev/null >> $O
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_abi_symbols.py -x -q 2>&1 | tail -3 >> $O
python - >> $O 2>&1 <<'PY'
import time, numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
w = workloads.didi_day("cfg2")
R = 1024
env = w.make_env(R)
t0 = time.perf_counter(); init = w.vehicle_nodes(R); t_gen = time.perf_counter() - t0
env.reset(init); env.sync()
def tm(f, n):
    f(); env.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    env.sync()
    return (time.perf_counter() - t0) / n * 1e3
seeds = (np.arange(R) + w.veh_seed).astype(np.uint64)
print("configs[1], 1024 replicas x 10k vehicles: native host generation of the start nodes %.1f ms; vds_reset (validation + capacity histogram + upload + reset) %.2f ms; "
      "vds_reset_random %.2f ms; vds_reset_again %.3f ms" % (t_gen * 1e3, tm(lambda: env.reset(init), 5), tm(lambda: env.reset_random(seeds), 20), tm(env.reset_again, 50)))
env.reset_random(seeds)
ok = all(np.array_equal(env.vehicles(r)["node"], init[r]) for r in (0, 1, 511, 1023))
print("same nodes as workloads.vehicle_nodes (replicas 0, 1, 511, 1023):", ok)
env.run(env.T); a = env.counters().copy()
env.reset(init); env.run(env.T); b = env.counters().copy()
print("same day:", np.array_equal(a, b))
PY
grep -v amdgpu $O
