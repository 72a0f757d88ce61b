"""configs[1] with D distinct order days: ms per day and slow-path buckets, dense layout vs wide (VDS_DENSE=0).
    python profiles/r04/probe_days.py [days] [replicas]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vehicles_dispatch_simulator_amd import workloads
D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = workloads.didi_day("cfg2")
init = w.vehicle_nodes(R)
days = workloads.distinct_days(w, D)
for label, rmap in (("interleaved", np.arange(R) % D), ("blocked", np.minimum(np.arange(R) // max(1, R // D), D - 1))):
    env = w.make_env(R, load=False)
    env.load_order_days(days, rmap.astype(np.int32))
    env.reset(init)
    T = env.T
    env.reset_again(); env.run(T); env.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        env.reset_again(); env.run(T)
    env.sync()
    dt = (time.perf_counter() - t0) / 5
    print(label, env.main_kernel(), "groups", env.run_groups(), "T", T, "ms/day %.3f" % (dt * 1e3), env.work())
    env.close()
