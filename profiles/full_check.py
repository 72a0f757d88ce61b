"""Full-scale parity: EVERY replica of the bench workload against the CPU oracle (per-order status / vehicle / wait
and counters, bit-exact).   python profiles/full_check.py [cfg2|cfg4] [replicas]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = workloads.didi_day("cfg2") if wl == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
init = w.vehicle_nodes(R)
env = w.make_env(R)
env.reset(init)
env.run(env.T)
cn = env.counters()
o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, w.release_min, w.pickup, w.delivery, w.vehicles)
t0 = time.time()
bad = 0
for r0 in range(0, R, 64):
    got = env.orders(r0, min(64, R - r0))
    for i in range(got["status"].shape[0]):
        r = r0 + i
        o.reset(init[r]); o.run_day()
        exp, oc = o.orders(), o.counters()
        ok = all(np.array_equal(got[k][i], exp[k]) for k in ("status", "vehicle", "wait")) and \
            (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
        bad += 0 if ok else 1
print("%s: %d replicas checked against the oracle in %.0f s, mismatching replicas: %d" % (wl, R, time.time() - t0, bad))
assert bad == 0
