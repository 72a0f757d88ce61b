"""Timing sweep of the lanes tick (k_tick_lanes) over its knobs, against the row-mapped kernel, on ONE box.

    python profiles/lanes_sweep.py [cfg2|cfg5] [days]
Each configuration runs in this process on a fresh handle (the knobs are read at vds_load_orders)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vehicles_dispatch_simulator_amd import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
days = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = workloads.didi_day("cfg2") if name == "cfg2" else workloads.stress()
R = 1024 if name == "cfg2" else 128
init = w.vehicle_nodes(R)
ref = None
configs = [("rows", 5, {})]
for spec in sys.argv[3:] or ["lg=-1,loc=32,keys=16", "lg=0,loc=128,keys=64", "lg=1,loc=64,keys=32", "lg=2,loc=32,keys=16", "lg=3,loc=32,keys=16", "lg=-1,loc=64,keys=32", "lg=-1,loc=48,keys=16"]:
    kv = dict(x.split("=") for x in spec.split(","))
    configs.append((spec, 6, kv))
for label, fg, kv in configs:
    for k in ("VDS_LANES_LG", "VDS_LANES_LOC", "VDS_LANES_KEYS"):
        os.environ.pop(k, None)
    if "lg" in kv and int(kv["lg"]) >= 0:
        os.environ["VDS_LANES_LG"] = kv["lg"]
    if "loc" in kv:
        os.environ["VDS_LANES_LOC"] = kv["loc"]
    if "keys" in kv:
        os.environ["VDS_LANES_KEYS"] = kv["keys"]
    env = w.make_env(R, force_generic=fg)
    env.reset(init)
    T = env.T
    env.reset_again(); env.run(T); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(days):
        env.reset_again(); env.run(T)
    env.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / days
    cn = env.counters()
    if ref is None:
        ref = cn
    print("%-28s %-13s %8.3f ms/day %7.1f us/tick  counters %s" % (label, env.main_kernel(), dt * 1e3, dt / T * 1e6, "==" if np.array_equal(cn, ref) else "DIFFER"), flush=True)
    env.close()
