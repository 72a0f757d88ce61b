import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
init = w.vehicle_nodes(1)
o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, w.release_min, w.pickup, w.delivery, w.vehicles)
o.reset(init[0])
T = o.num_ticks
prev = o.counters()
rows = []
for t in range(T):
    o.begin_tick()
    ob = o.obs(); cn = o.counters()
    orders = cn["order_num"] - prev["order_num"]; rej = cn["reject_num"] - prev["reject_num"]
    pre, clo = np.asarray(ob["idle_pre"]), np.asarray(ob["cl_orders"])
    dry = int(np.maximum(clo - pre, 0).sum()); own = int(np.minimum(clo, pre).sum())
    rows.append((t, int(orders), int(rej), dry, int(orders - rej) - own, int(pre.sum()), int((clo > pre).sum())))
    prev = cn
    o.end_tick()
print("tick orders rejects dry(approx) matched_by_neighbour(approx) idle_before dry_clusters")
for r in rows: print("%3d %5d %5d %5d %5d %6d %4d" % r)
