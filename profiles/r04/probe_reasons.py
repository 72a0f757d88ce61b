"""Why buckets leave k_tick_dense's fast path, and how often the wide arrival-ranking variants run (instrumented build):
    VDS_LIB=$PWD/build/libvds_prof.so VDS_DENSE_LPR=8|16 python profiles/r04/probe_reasons.py D [replicas]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vehicles_dispatch_simulator_amd import workloads
D = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = workloads.didi_day("cfg2")
env = w.make_env(R, load=False)
if D > 1:
    env.load_order_days(workloads.distinct_days(w, D), (np.arange(R) % D).astype(np.int32))
else:
    env.load_orders(w.release_min, w.pickup, w.delivery)
env.reset(w.vehicle_nodes(R))
env._lib.vds_debug_ablate(env._h, 2097152)          # count the wavefronts of the fast body as well
env.run(env.T); env.sync()
e = np.zeros(16, dtype=np.int32)
env._lib.vds_debug_read_err(env._h, e.ctypes.data)
names = {4: "k > 64 orders", 5: "n > 128 candidates", 6: "far entries", 7: "ring entries > keys", 8: "arrivals > 128", 9: "list + arrivals > 128 (or idle_cap)", 10: "slot arrivals > 2 x keys",
         11: "candidates + ring > 128", 12: "wavefronts in the fast body", 13: "... ranking > 4 x LPR slots", 14: "... ranking > 8 x LPR slots", 15: "... 128-entry tables"}
print("D %d lpr %s %s slow buckets %d of %d bucket-ticks" % (D, os.environ.get("VDS_DENSE_LPR", "default"), env.main_kernel(), e[2], env.T * R * env.C))
for i in range(4, 16):
    print("  %-40s %10d" % (names[i], e[i]))
env.close()
