"""Event counts of k_dfs_walk_gs' deferred-acceptance phase (make -C vehicles_dispatch_simulator_amd/csrc variant NAME=gss EXTRA=-DGS_STATS;
VDS_LIB=build/libvds_gss.so python profiles/gs_stats.py [replicas] [first tick] [ticks])"""
import sys
sys.path.insert(0, ".")
import numpy as np
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
env = w.make_env(R)
env.reset(w.vehicle_nodes(R))
print(env.main_kernel())
n = n or env.T - t0
if t0: env.run(t0)
env.sync()
err = np.zeros(16, dtype=np.int32)
env._lib.vds_debug_read_err(env._h, err.ctypes.data)
base = err.copy()
env.run(n); env.sync()
env._lib.vds_debug_read_err(env._h, err.ctypes.data)
d = (err - base).astype(np.float64)
rt = min(R, 16) * n          # (the instrumented build counts the first 16 replicas of a launch)
names = ["batches", "orders scanned", "re-scan requests", "lost cas", "dry holders displaced", "re-pick chains", "chain steps", "no candidate"]
print("ticks %d..%d" % (t0, t0 + n))
for i, nm in enumerate(names): print("%-24s %8.2f per replica-tick" % (nm, d[4 + i] / rt))
for i, nm in enumerate(["scanning", "proposing", "waiting", "claiming"]): print("%-24s %8.1f kilo-cycles per wavefront and tick" % (nm, d[12 + i] / rt / 4))
if d[4]: print("orders per batch %.2f, kilo-cycles per batch: scan %.1f, proposals %.1f" % (d[5] / d[4], d[12] / d[4], d[13] / d[4]))
