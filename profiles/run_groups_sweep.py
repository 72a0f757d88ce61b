"""vds_run with the replicas as independent groups (parallel branches of the day graph, vds_set_run_groups) - timing sweep on
one box; the totals printed behind every line must not change.

    python profiles/run_groups_sweep.py [R] [days] ["G:stagger G:stagger ..."] [cfg4|cfg2]
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vehicles_dispatch_simulator_amd import workloads

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
days = int(sys.argv[2]) if len(sys.argv) > 2 else 20
combos = sys.argv[3].split() if len(sys.argv) > 3 else ["1:0", "2:0", "2:1", "2:2", "3:2", "4:2", "4:0", "6:2", "8:2"]
name = sys.argv[4] if len(sys.argv) > 4 else "cfg4"
w = workloads.didi_day("cfg2") if name == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
e = w.make_env(R)
e.reset(w.vehicle_nodes(R))
T = e.T
ref = None
for cb in combos:
    G, st = (int(x) for x in cb.split(":"))
    e.set_run_groups(G, st)
    for _ in range(2):
        e.reset_again(); e.run(T)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(days):
        e.reset_again(); e.run(T)
    e.sync()
    dt = time.perf_counter() - t0
    tot = e.total_counters()
    print("R %d groups %d stagger %d  ms/day %.3f  us/tick %.1f -> %.4g env-steps*replicas/s  %s" % (
        R, G, st, dt / days * 1e3, dt / days / T * 1e6, T * R * days / dt, "" if tot is None else list(map(int, tot))), flush=True)
