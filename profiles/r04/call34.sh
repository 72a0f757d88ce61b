#!/bin/bash
# vds_reset with the start-node checks on the device: tests, timing
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c34.txt; : > $O
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())" 2>/dev/null >> $O
timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_simulation_shell.py tests/test_gpu_mutable_surface.py tests/test_gpu_replica_days.py -x -q 2>&1 | tail -3 >> $O
python - >> $O 2>&1 <<'PY'
import os, time, numpy as np
from vehicles_dispatch_simulator_amd import workloads
w = workloads.didi_day("cfg2")
R = 1024
init = w.vehicle_nodes(R)
def tm(env, f, n):
    f(); env.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    env.sync()
    return (time.perf_counter() - t0) / n * 1e3
for hc in ("1", "0"):
    os.environ["VDS_RESET_HOST_CHECKS"] = hc
    env = w.make_env(R)
    env.reset(init)
    print("VDS_RESET_HOST_CHECKS=%s: vds_reset of 1024 x 10k start nodes %.2f ms" % (hc, tm(env, lambda: env.reset(init), 10)))
    env.close()
PY
grep -v amdgpu $O
