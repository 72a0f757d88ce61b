"""Per-section cycle attribution of k_tick_rows (s_memtime stamps with s_waitcnt 0 at the boundaries;
perturbs overlap, gives attribution):   python profiles/sections.py [replicas]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = workloads.didi_day("cfg2")
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
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
names = ["0 header loads + LDS staging + barrier", "1 issue idle loads (wait)", "2 ring loads + ranking", "3 merge arrivals / setup", "4 match loop", "5 results + posts", "6 compaction + hdr/cnt stores"]
waves = int(buf[15])
tot = float(buf[:7].sum())
print("instrumented kernel: %.1f us/launch, %d waves reached the end" % (ms.mean() * 1e3, waves))
for i, n in enumerate(names):
    print("%-42s %8.0f cycles/wave  %5.1f%%" % (n, buf[i] / max(waves, 1), 100.0 * buf[i] / tot))
print("total %.0f ticks/wave/day -> %.0f ticks per wave-launch (s_memtime ticks are 100 MHz = 10 ns)" % (tot / max(waves, 1), tot / max(waves, 1) / T))
