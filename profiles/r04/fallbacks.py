"""What the documented fallbacks of the plain tick cost at configs[1] ("limits pick slow paths - none has a
throughput number"): the default (k_tick_dense, static arrival slots), the dense tick with the arrival ring (what it takes when an
order's arrival can lie more than DENSE_PULL_WMAX slots behind its earliest slot), the wide layout (k_tick_rows: costs beyond the
dense keys, clusters of more than 255 nodes, V >= 2^24), the generic kernel (k_tick: costs >= 2^23 or above the pickup window), and
the dense tick when every bucket takes its slow path.     python profiles/r04/fallbacks.py [replicas] [days]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
days = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = workloads.didi_day("cfg2")
init = w.vehicle_nodes(R)
for label, kw, nd in (("default: k_tick_dense, static arrival slots", {}, days),
                      ("dense tick with the arrival ring (dense_debug force_slow bit 1)", {"dense_debug": (0, 0, 0, 2)}, days),
                      ("wide layout, k_tick_rows (force_generic 5)", {"force_generic": 5}, days),
                      ("generic kernel k_tick, one wavefront per bucket (force_generic 1)", {"force_generic": 1}, max(2, days // 4)),
                      ("dense tick, EVERY bucket on its slow path (dense_debug force_slow bit 0)", {"dense_debug": (0, 0, 0, 1)}, 2)):
    env = w.make_env(R, **kw)
    env.reset(init)
    T = env.T
    env.run(T); env.sync()
    t0 = time.perf_counter()
    for _ in range(nd):
        env.reset_again(); env.run(T)
    env.sync()
    dt = (time.perf_counter() - t0) / nd
    print("%-80s %-14s %9.3f ms per day  %.3e env-steps*replicas/s  slow-path buckets %d" % (label, env.main_kernel(), dt * 1e3, T * R / dt, env.work()["slow_path_buckets"]), flush=True)
    env.close()
