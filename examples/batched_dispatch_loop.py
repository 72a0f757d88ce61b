"""A vectorised dispatch loop over R city replicas: observations stay on the GPU as a torch tensor, a toy policy
moves idle vehicles from the clusters with the largest surplus towards the clusters with the largest expected
shortage, every replica with its own vehicle seed.

    python examples/batched_dispatch_loop.py [replicas]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vehicles_dispatch_simulator_amd import workloads

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = workloads.tiny(N=600, C=24, vehicles=400, orders=12000)
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
some_node_of = np.array([np.flatnonzero(w.city.node2cluster == c)[0] for c in range(w.city.C)], dtype=np.int32)
t0 = time.time()
for t in range(env.T):
    env.step()
    obs = env.obs_torch()                                   # int32 [5, R, C] on the GPU, zero copy
    idle, supply, demand = obs[1].float(), obs[2].float(), obs[3].float()
    surplus = idle + supply - demand                        # [R, C]
    src = surplus.argmax(dim=1)                             # richest cluster of every replica
    dst = surplus.argmin(dim=1)                             # poorest cluster of every replica
    move = (surplus.gather(1, src[:, None])[:, 0] - surplus.gather(1, dst[:, None])[:, 0] > 4) & (idle.gather(1, src[:, None])[:, 0] > 0)
    rep = torch.nonzero(move)[:, 0].cpu().numpy().astype(np.int32)
    if rep.size:
        s, d = src.cpu().numpy()[rep], dst.cpu().numpy()[rep]
        env.apply_dispatch(rep, s.astype(np.int32), np.zeros(rep.size, np.int32), some_node_of[d])   # first idle vehicle of src
    env.advance()
env.sync()
cn = env.counters()
print("replicas %d, %d ticks in %.2f s; rejects per replica: mean %.1f (min %d, max %d); dispatches %d" % (
    R, env.T, time.time() - t0, cn[:, 1].mean(), cn[:, 1].min(), cn[:, 1].max(), int(cn[:, 4].sum())))
