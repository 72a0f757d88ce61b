#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python - 2>&1 <<'PY' | grep -v amdgpu > gpurun_out/r04_c37.txt
import time, numpy as np
from vehicles_dispatch_simulator_amd import workloads
for name in ("cfg2", "cfg4"):
    w = workloads.didi_day("cfg2") if name == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
    t0 = time.perf_counter(); env = w.make_env(1024, load=False); t1 = time.perf_counter()
    env.load_orders(w.release_min, w.pickup, w.delivery); env.sync(); t2 = time.perf_counter()
    env.load_orders(w.release_min, w.pickup, w.delivery); env.sync(); t3 = time.perf_counter()
    days = workloads.distinct_days(w, 16)
    t4 = time.perf_counter(); env.load_order_days(days, (np.arange(1024) % 16).astype(np.int32)); env.sync(); t5 = time.perf_counter()
    print("%s: create (static tables) %.0f ms, vds_load_orders first %.0f ms, again (Reload) %.0f ms, vds_load_order_days(16 days) %.0f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3))
    env.close()
PY
cat gpurun_out/r04_c37.txt
