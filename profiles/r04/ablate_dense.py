"""Timing-only ablation of k_tick_dense (instrumented build: make prof, VDS_LIB=build/libvds_prof.so; results invalid for
non-zero flags):   VDS_LIB=build/libvds_prof.so python profiles/r04/ablate_dense.py [replicas]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = workloads.didi_day("cfg2")
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
print(env.main_kernel(), env._lib.vds_build_id().decode())
# state-preserving switches first (the day evolves as usual, timings are comparable); the others change what the later slots see
names = [(0, "full"), (4096, "+ every ring post twice (shadow table)"), (8192, "+ the ring post's atomic twice"), (16384, "+ the ring post's entry store twice"),
         (65536, "+ the arrival-slot store twice"), (131072 | 32768, "no counter loads / stores"),
         (2, "no idle write-back (counts kept)"), (8, "no result stores"), (32768, "no counter stores"), (2 | 8 | 32768, "no write-back / result / counter stores"),
         (0, "full (again)"),
         (1, "[state-changing] no posts"), (4 | 1 | 2 | 8 | 16, "[state-changing] no match loop, no stores"),
         (2048 | 4 | 1 | 2 | 8 | 16 | 1024, "[state-changing] header loads + order records + control flow only"), (256, "header loads only"), (512, "empty kernel")]
for f, nm in names:
    env._lib.vds_debug_ablate(env._h, f)
    best = None
    for rep in range(3):
        env.reset_again(); env.profile(True); env.run(T); ms = env.profile_read(T + 8); env.profile(False)
        try:
            env.sync()
        except Exception:
            pass
        best = float(ms.mean()) if best is None else min(best, float(ms.mean()))
    print("%-52s %7.1f us/launch" % (nm, best * 1e3), flush=True)
env._lib.vds_debug_ablate(env._h, 0)
