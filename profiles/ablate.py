"""Timing-only ablation of the fast tick kernel (results invalid for non-zero flags):
    python profiles/ablate.py [replicas]"""
import sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = workloads.didi_day("cfg2")
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
names = {0: "full", 1: "no posts (atomics+ring stores)", 32: "posts: atomic only (add 0)", 64: "posts: entry store only", 2: "no idle write-back", 8: "no result stores"}
out = {}
for f, nm in names.items():
    env._lib.vds_debug_ablate(env._h, f)
    best = None
    for rep in range(2):
        env.reset_again(); env.profile(True); env.run(T); ms = env.profile_read(T + 8); env.profile(False)
        try:
            env.sync()
        except Exception as e:
            pass
        best = float(ms.mean()) if best is None else min(best, float(ms.mean()))
    out[nm] = best
    print("%-34s %.1f us/launch" % (nm, best * 1e3), flush=True)
env._lib.vds_debug_ablate(env._h, 0)
