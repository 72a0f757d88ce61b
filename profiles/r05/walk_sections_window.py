"""k_dfs_walk's section attribution over a WINDOW of ticks (instrumented build): where a heavy slot's time goes against a light one's.
   VDS_LIB=$PWD/build/libvds_prof.so VDS_RUN_GROUPS=1 python profiles/r05/walk_sections_window.py"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = 1024
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
init = w.vehicle_nodes(R)
sec = [("tables (ranks, lengths) + barrier", 24), ("prefix of the list lengths + barrier", 25), ("stamps from the preliminary results", 26), ("dry bits + barrier", 0),
       ("the walk (wavefront 0 serves, 1-3 scan)", 1), ("steal log -> results, tk", 29), ("evaluations (a): dry orders' visit rows", 30),
       ("evaluations (b): steals per searching cluster", 31), ("reduce + closed form + barrier", 5), ("resolve: vehicle ids, arrivals, counters", 27),
       ("compaction, lists up to 64", 28), ("compaction, longer lists", 7)]
for label, t0, n in (("clean slots 2-7", 2, 6), ("first busy hour 49-54", 49, 6), ("heavy slots 79-84", 79, 6), ("heavy slots 109-114", 109, 6)):
    env.reset(init)
    env._lib.vds_debug_ablate(env._h, 0)
    env.run(t0); env.sync()
    a = np.zeros(32, dtype=np.uint64); env._lib.vds_debug_read_prof(env._h, a.ctypes.data)
    env._lib.vds_debug_ablate(env._h, 128)
    env.run(n); env.sync()
    b = np.zeros(32, dtype=np.uint64); env._lib.vds_debug_read_prof(env._h, b.ctypes.data)
    env._lib.vds_debug_ablate(env._h, 0)
    buf = (b - a).astype(np.float64) if (b >= a).all() else b.astype(np.float64)
    T = n
    print("==== %s (%s)" % (label, env.main_kernel()))
    print("dry orders per replica-tick %.1f, served by a neighbour %.1f, with a redo chain %.2f, scanned (again) by the walk itself %.2f" % (buf[6] / R / T, buf[2] / R / T, buf[3] / R / T, buf[4] / R / T))
    print("wavefront 0 per replica-tick: %.0f cycles waiting for a record, %.0f in redo chains; scanning wavefronts: %.1f scans of %.0f cycles" % (buf[8] / R / T, buf[9] / R / T, buf[11] / R / T, buf[10] / max(1, buf[11])))
    print("wavefront 0 per served order: record + candidate stamps %.0f cycles, winner + steal (without the chain) %.0f, next dry order %.0f" % (buf[12] / max(1, buf[6]), buf[13] / max(1, buf[6]), buf[14] / max(1, buf[6])))
    sc = [float(buf[16 + i]) / max(1, buf[11]) for i in range(5)]
    print("one scan: visit row + bounds %.0f, alive counts %.0f, first pass %.0f, second pass %.0f, the K best %.0f cycles; %.2f eight-slot groups" % (sc[0], sc[1], sc[2], sc[3], sc[4], float(buf[21]) / max(1, buf[11])))
    tot = sum(float(buf[i]) for _, i in sec)
    print("total %.0f cycles per wavefront and tick" % (tot / (R * 4) / T))
    for nm, i in sec:
        print("  %-50s %9.0f cycles/wave/tick  %5.1f%%" % (nm, buf[i] / (R * 4) / T, 100.0 * buf[i] / tot))
env.close()
