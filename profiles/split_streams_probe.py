"""Probe (round 3): does the hybrid neighbour-search tick gain from running G groups of replicas on G streams (k_tick_rows of
one group under k_dfs_walk of another)?  G separate handles of R / G replicas each, every handle on its own stream.

    python profiles/split_streams_probe.py [cfg4|cfg2] [R] [days]
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vehicles_dispatch_simulator_amd import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
days = int(sys.argv[3]) if len(sys.argv) > 3 else 20
w = workloads.didi_day("cfg2") if name == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
for G in (1, 2, 4, 8):
    envs = [w.make_env(R // G) for _ in range(G)]
    nodes = w.vehicle_nodes(R)
    for g, e in enumerate(envs):
        e.reset(nodes[g * (R // G):(g + 1) * (R // G)])
    T = envs[0].T
    for _ in range(2):
        for e in envs: e.reset_again(); e.run(T)
    for e in envs: e.sync()
    t0 = time.perf_counter()
    for _ in range(days):
        for e in envs: e.reset_again(); e.run(T)
    for e in envs: e.sync()
    dt = time.perf_counter() - t0
    print("G %d  %s  ms/day %.3f  -> %.4g env-steps*replicas/s" % (G, envs[0].main_kernel(), dt / days * 1e3, T * R * days / dt), flush=True)
    for e in envs: e.close()
