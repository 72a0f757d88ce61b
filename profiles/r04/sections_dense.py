"""Per-section attribution of k_tick_dense (instrumented build; s_memtime stamps, s_waitcnt 0 at most boundaries: perturbs the
overlap, gives attribution):   VDS_LIB=$PWD/build/libvds_prof.so VDS_RUN_GRAPH=0 python profiles/r04/sections_dense.py [replicas [order days]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # order days (replica r on day r % D)
w = workloads.didi_day("cfg2")
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream, load=D <= 1)
if D > 1:
    env.load_order_days(workloads.distinct_days(w, D), (np.arange(R) % D).astype(np.int32))
env.reset(w.vehicle_nodes(R))
T = env.T
env.run(T); env.sync()
env.reset_again()
buf = np.zeros(32, dtype=np.uint64)
env._lib.vds_debug_read_prof(env._h, buf.ctypes.data)
env._lib.vds_debug_ablate(env._h, 128)
env.profile(True); env.run(T); ms = env.profile_read(T + 8); env.profile(False)
env._lib.vds_debug_read_prof(env._h, buf.ctypes.data)
env._lib.vds_debug_ablate(env._h, 0)
names = ["0 scalar loads, header words, candidate entries, staging loads arrived", "1 barrier", "2 arrival count, row classification", "3 idle list chunk loaded",
         "4 arrivals ranked + merged through the table", "5 loc bytes", "6 match loop", "7 compaction + write-back", "8 results, arrival slots, header, counters"]
waves = int(buf[15])
tot = float(buf[:9].sum())
print(env.main_kernel(), env._lib.vds_build_id().decode(), "order days", D)
print("instrumented kernel: %.1f us/launch, %d wave-launches recorded section 6" % (ms.mean() * 1e3, waves))
for i, n in enumerate(names):
    print("%-74s %8.1f ticks/wave  %5.1f%%" % (n, buf[i] / max(waves, 1), 100.0 * buf[i] / tot))
print("total %.1f s_memtime ticks per wave-launch (100 MHz: 10 ns each)" % (tot / max(waves, 1)))
