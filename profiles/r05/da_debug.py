"""Tick-by-tick comparison of the hybrid tick's deferred-acceptance form with the oracle on a tiny golden city (debugging aid).
   VDS_WALK_DA=1 python profiles/r05/da_debug.py [fixture] [replicas]"""
import os, sys
os.environ.setdefault("VDS_WALK_DA", "1")
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from helpers import load_golden, make_oracle, engine_settings
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_kmeans_dfs2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = load_golden(name)
env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=int(g["V"]), depth_limit=int(g["depth_limit"]),
                         neighbor_can_server=bool(g["neighbor_can_server"]), **engine_settings(g))
env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
print(env.main_kernel(), "T", env.T, flush=True)
env.reset(np.tile(g["veh_node"], (R, 1)))
o = make_oracle(g)
for t in range(env.T):
    env.step(); env.sync(); o.begin_tick()
    got, exp = env.orders(0, 1), o.orders()
    bad = [k for k in ("status", "vehicle", "wait") if not np.array_equal(got[k][0], exp[k])]
    cn, oc = env.counters(), o.counters()
    if bad or cn[0, 7] != oc["evals"] or cn[0, 1] != oc["reject_num"]:
        d = np.flatnonzero((got["status"][0] != exp["status"]) | (got["vehicle"][0] != exp["vehicle"]) | (got["wait"][0] != exp["wait"]))
        print("tick", t, "differs:", bad, "orders", d[:12].tolist(), "evals", cn[0, 7], oc["evals"], "rejects", cn[0, 1], oc["reject_num"])
        for i in d[:8]:
            print("  order", i, "pickup cluster", int(g["node2cluster"][g["o_pickup"][i]]), "got", got["status"][0][i], got["vehicle"][0][i], got["wait"][0][i], "exp", exp["status"][i], exp["vehicle"][i], exp["wait"][i])
        break
    env.advance(); o.end_tick()
else:
    print("all", env.T, "ticks equal")
env.close()
