"""The neighbour-search tick per slot at small and full replica counts (profile mode: events around the main kernel of each tick,
which in the neighbour-search form is the walk): what of a slot is the per-replica chain and what is load."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
res = {}
for R in [int(x) for x in sys.argv[1:]] or [32, 1024]:
    env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
    env.reset(w.vehicle_nodes(R)); env.run(env.T); env.reset_again()
    env.profile(True); env.run(env.T); ms = env.profile_read(env.T + 8); env.profile(False)
    res[R] = ms[-env.T:] * 1e3
    print("R=%d main kernel %s: day %.2f ms" % (R, env.main_kernel(), res[R].sum() / 1e3))
Rs = sorted(res)
for t in range(0, 148, 8):
    print("%3d " % t + "  ".join("/".join("%4.0f" % res[R][t + i] for R in Rs) for i in range(min(8, 148 - t))))
