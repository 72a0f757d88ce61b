"""One hooked day (step -> k_pack_obs -> k_dispatch_dense with a fixed action tensor -> advance) for a kernel trace:
    rocprofv3 --kernel-trace --stats -- python profiles/r04/hooked_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R, K = 1024, 8
w = workloads.didi_day("cfg2")
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
n2c = np.asarray(w.city.node2cluster)
node_of = torch.tensor([int(np.flatnonzero(n2c == c)[0]) for c in range(env.C)], dtype=torch.int32, device="cuda")
act = torch.full((R, K, 3), -1, dtype=torch.int32, device="cuda")
act[:, :, 1] = 0
act[:, :, 2] = node_of[:K][None, :]
ar = torch.arange(R, device="cuda", dtype=torch.int32)
act[:, 0, 0] = ar % env.C
act[:, 0, 2] = node_of[((ar + 97) % env.C).long()]
for day in range(2):
    env.reset_again()
    for _ in range(env.T):
        env.step(); env.obs_torch(inflight=False); env.apply_dispatch_torch(act); env.advance()
torch.cuda.synchronize()
try:
    env.sync()
except Exception as e:
    print(e)
print(env.work())
