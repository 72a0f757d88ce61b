import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from helpers import load_golden
from vehicles_dispatch_simulator_amd import workloads, BatchedDispatchEnv
import test_gpu_parity as tp
g = load_golden("tiny_kmeans")
for rep in range(3):
    t0 = time.perf_counter(); env = tp.make_env(g, 5); t1 = time.perf_counter()
    env.reset(np.tile(g["veh_node"], (5, 1))); t2 = time.perf_counter()
    for t in range(env.T): env.step(); env.advance()
    env.sync(); t3 = time.perf_counter()
    od = env.orders(); t4 = time.perf_counter()
    env.close(); t5 = time.perf_counter()
    print("make_env %.1f ms  reset %.1f  %d steps %.1f  orders %.1f  close %.1f" % ((t1-t0)*1e3, (t2-t1)*1e3, env.T, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3))
import os
os.environ["VDS_LOAD_TIMING"] = "1"
env = tp.make_env(g, 5); env.close()
