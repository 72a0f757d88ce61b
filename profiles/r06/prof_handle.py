import time, sys, numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import load_golden
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv
g = load_golden("tiny_kmeans_dfs2")
import cProfile, pstats
def one():
    t0=time.perf_counter()
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=3, vehicles=int(g["V"]), depth_limit=int(g["depth_limit"]), neighbor_can_server=True)
    t1=time.perf_counter()
    env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
    t2=time.perf_counter()
    env.reset(np.tile(g["veh_node"], (3,1)).astype(np.int32))
    t3=time.perf_counter()
    env.run(env.T); env.sync()
    t4=time.perf_counter()
    env.close()
    t5=time.perf_counter()
    return [t1-t0,t2-t1,t3-t2,t4-t3,t5-t4]
one()
r=np.array([one() for _ in range(10)])
print("create %.1f load_orders %.1f reset %.1f run %.1f close %.1f ms" % tuple(r.mean(0)*1e3))
