"""Per-section cycle attribution of the neighbour-search tick's second kernel (instrumented build: make -C vehicles_dispatch_simulator_amd/csrc prof):
VDS_LIB=/root/repo/build/libvds_prof.so python profiles/sections_dfs.py [replicas] [first slot] [slots]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T = int(sys.argv[3]) if len(sys.argv) > 3 else T - t0
if t0:
    env.run(t0)
env.sync()
print("slots %d .. %d" % (t0, t0 + T))
buf = np.zeros(32, dtype=np.uint64)
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
if env.main_kernel() in ("k_dfs_hybrid", "k_dfs_dense"):
    # k_dfs_walk, 4 wavefronts per replica: cycles per wavefront and tick between the stamps (all wavefronts wait at the barriers)
    sec = [("tables (ranks, lengths) + barrier", 24), ("prefix of the list lengths + barrier", 25), ("stamps from the preliminary results", 26), ("dry bits + barrier", 0),
           ("the walk (wavefront 0 serves, 1-3 scan)", 1), ("steal log -> results, tk", 29), ("evaluations (a): dry orders' visit rows", 30),
           ("evaluations (b): steals per searching cluster", 31), ("reduce + closed form + barrier", 5), ("resolve: vehicle ids, arrivals, counters", 27),
           ("compaction, lists up to 64", 28), ("compaction, longer lists", 7)]
    print("dry orders per replica-tick %.1f, served by a neighbour %.1f, with a redo chain %.2f, scanned (again) by the walk itself %.2f" % (
        buf[6] / R / T, buf[2] / R / T, buf[3] / R / T, buf[4] / R / T))
    print("wavefront 0 per replica-tick: %.0f cycles waiting for a record, %.0f in redo chains; scanning wavefronts: %.1f scans of %.0f cycles" % (
        buf[8] / R / T, buf[9] / R / T, buf[11] / R / T, buf[10] / max(1, buf[11])))
    print("wavefront 0 per served order: record + candidate stamps %.0f cycles, winner + steal (without the chain) %.0f, next dry order %.0f" % (
        buf[12] / max(1, buf[6]), buf[13] / max(1, buf[6]), buf[14] / max(1, buf[6])))
    sc = [float(buf[16 + i]) / max(1, buf[11]) for i in range(5)]
    print("one scan: visit row + bounds %.0f, alive counts %.0f, first pass %.0f, second pass %.0f, the %s best %.0f cycles; %.2f eight-slot groups; %.2f of the scans serve two orders of one bucket" % (
        sc[0], sc[1], sc[2], sc[3], "WK_K", sc[4], float(buf[21]) / max(1, buf[11]), float(buf[22]) / max(1, buf[11])))
    print("searching clusters with dry orders per replica-tick: %.1f" % (float(buf[23]) / R / T))
    print("instrumented: %.2f ms/launch" % ms.mean())
    tot = sum(float(buf[i]) for _, i in sec)
    for n, i in sec:
        print("%-50s %9.0f cycles/wave/tick  %5.1f%%" % (n, buf[i] / (R * 4) / T, 100.0 * buf[i] / tot))
    sys.exit(0)
print(env.main_kernel())
tot = float(buf[:6].sum() + buf[7])
print("instrumented: %.2f ms/launch; DFS rounds per replica-tick: %.1f" % (ms.mean(), buf[6] / nw / T))
for i, n in enumerate(names):
    if n is None:
        continue
    print("%-40s %9.0f cycles/wave/tick  %5.1f%%" % (n, buf[i] / nw / T, 100.0 * buf[i] / tot))
