"""Where vds_load_orders* spends its time on a loaded handle (Reload, simulator.py:130-212):  VDS_LOAD_TIMING=1 python profiles/r05/load_timing.py [cfg2|cfg4]"""
import os, sys, time
os.environ["VDS_LOAD_TIMING"] = "1"
sys.path.insert(0, ".")
import numpy as np
from vehicles_dispatch_simulator_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = workloads.didi_day("cfg2") if wl == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
R = 1024
t0 = time.perf_counter(); env = w.make_env(R, load=False); print("create + static tables %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for i in range(3):
    print("---- load %d" % i, file=sys.stderr, flush=True)
    t0 = time.perf_counter(); env.load_orders(w.release_min, w.pickup, w.delivery); print("vds_load_orders #%d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
env.reset(w.vehicle_nodes(R)); env.run(env.T); env.sync()
days = workloads.distinct_days(w, 16)
print("---- 16 days", file=sys.stderr, flush=True)
for i in range(3):
    print("---- 16 days, load %d" % i, file=sys.stderr, flush=True)
    t0 = time.perf_counter(); env.load_order_days(days, (np.arange(R) % 16).astype(np.int32)); print("vds_load_order_days(16) #%d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
env.close()
