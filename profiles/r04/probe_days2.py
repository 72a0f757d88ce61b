"""configs[1] with D distinct order days, replica r on day r % D (D <= 64: stored regrouped, day mode 1; larger: one order stream per
row, day mode 2): ms per day, slow-path buckets, kernel.   VDS_DENSE_LPR=8|16 python profiles/r04/probe_days2.py D [replicas]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vehicles_dispatch_simulator_amd import workloads
D = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = workloads.didi_day("cfg2")
init = w.vehicle_nodes(R)
t0 = time.perf_counter(); days = workloads.distinct_days(w, D); tg = time.perf_counter() - t0
env = w.make_env(R, load=False)
t0 = time.perf_counter(); env.load_order_days(days, (np.arange(R) % D).astype(np.int32)); tl = time.perf_counter() - t0
env.reset(init)
T = env.T
env.reset_again(); env.run(T); env.sync()
t0 = time.perf_counter()
for _ in range(5):
    env.reset_again(); env.run(T)
env.sync()
dt = (time.perf_counter() - t0) / 5
wk = env.work()
print("D %d lpr %s %s groups %d T %d  ms/day %.3f  %.3e env-steps*replicas/s  slow buckets %d  evals %d  (days generated in %.1f s, loaded in %.1f s)" % (
    D, os.environ.get("VDS_DENSE_LPR", "default"), env.main_kernel(), env.run_groups(), T, dt * 1e3, T * R / dt, wk["slow_path_buckets"], wk["evals"], tg, tl), flush=True)
env.close()
