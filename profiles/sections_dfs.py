"""Per-section cycle attribution of k_tick_replica (instrumented build):  VDS_LIB=libvds_prof.so python profiles/sections_dfs.py [replicas]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
buf = np.zeros(16, dtype=np.uint64)
env._lib.vds_debug_read_prof(env._h, buf.ctypes.data)
env._lib.vds_debug_ablate(env._h, 128)
env.profile(True); env.run(T); ms = env.profile_read(T + 8); env.profile(False)
env._lib.vds_debug_read_prof(env._h, buf.ctypes.data)
env._lib.vds_debug_ablate(env._h, 0)
names = ["0 update (drain + hdr)", "1 node mirror build", "2 LB reduction", "3 bucket-parallel own-cluster match", "4 DFS candidate scan", "5 DFS winner (+ removal + post in v1)",
         None, "7 resolve results + compaction + flush (v2)"]
nw = R * 4
if env.main_kernel() == "k_tick_replica3":
    names = ["0 update (drain + hdr)", "1 mirror build", "2 phase 1: own-cluster pass, every bucket once", "3 dry bitset", "4 per dry order: pick + row + candidate scan",
             "5 per dry order: barrier + winner (+ redo) + barrier", None, "7 evaluations + resolve + compaction + flush"]
    nw = R * 8
if env.main_kernel() == "k_dfs_hybrid":
    names = ["0 k_dfs_walk: tables + stamps into LDS + dry bits", "1 k_dfs_walk: the walk (wavefront 0)", None, None, None, "5 k_dfs_walk: evaluations from the stamps", None, "7 k_dfs_walk: resolve + compaction"]
    nw = R * 4
    print("dry orders per replica-tick %.1f, served by a neighbour %.1f, with a redo chain %.2f, scanned again by the walk %.2f" % (
        buf[6] / R / T, buf[2] / R / T, buf[3] / R / T, buf[4] / R / T))
    print("wavefront 0 per replica-tick: %.0f cycles waiting for a record, %.0f in redo chains; scanning wavefronts: %.1f scans of %.0f cycles" % (
        buf[8] / R / T, buf[9] / R / T, buf[11] / R / T, buf[10] / max(1, buf[11])))
    print("wavefront 0 per served order: record + alive counts + candidate stamps %.0f cycles, winner + steal %.0f, result + next dry order %.0f" % (
        buf[12] / max(1, buf[6]), buf[13] / max(1, buf[6]), buf[14] / max(1, buf[6])))
    buf[2] = buf[3] = buf[4] = 0
print(env.main_kernel())
tot = float(buf[:6].sum() + buf[7])
print("instrumented: %.2f ms/launch; DFS rounds per replica-tick: %.1f" % (ms.mean(), buf[6] / nw / T))
for i, n in enumerate(names):
    if n is None:
        continue
    print("%-40s %9.0f cycles/wave/tick  %5.1f%%" % (n, buf[i] / nw / T, 100.0 * buf[i] / tot))
