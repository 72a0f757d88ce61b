"""A vectorised dispatch loop over R city replicas: observations stay on the GPU as a torch tensor, a toy policy
moves idle vehicles from the clusters with the largest surplus towards the clusters with the largest expected
shortage, every replica with its own vehicle seed.  Observation -> policy -> action never leaves the device.

    python examples/batched_dispatch_loop.py [replicas] [tiny|cfg2]        (cfg2 = the 192-cluster / 10k-vehicle bench city)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vehicles_dispatch_simulator_amd import workloads

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = workloads.didi_day("cfg2") if (len(sys.argv) > 2 and sys.argv[2] == "cfg2") else workloads.tiny(N=600, C=24, vehicles=400, orders=12000)
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
# start nodes drawn on the GPU: replica r as random.Random(w.veh_seed + r) would draw them (= env.reset(w.vehicle_nodes(R)), bit for bit)
env.reset_random(np.arange(R, dtype=np.uint64) + np.uint64(w.veh_seed))
some_node_of = torch.tensor([int(np.flatnonzero(w.city.node2cluster == c)[0]) for c in range(w.city.C)], dtype=torch.int32, device="cuda")
t0 = time.time()
for t in range(env.T):
    env.step()
    obs = env.obs_torch()                                   # int32 [5, R, C] on the GPU, zero copy
    idle, supply, demand = obs[1].float(), obs[2].float(), obs[3].float()
    surplus = idle + supply - demand                        # [R, C]
    src = surplus.argmax(dim=1)                             # richest cluster of every replica
    dst = surplus.argmin(dim=1)                             # poorest cluster of every replica
    move = (surplus.gather(1, src[:, None])[:, 0] - surplus.gather(1, dst[:, None])[:, 0] > 4) & (idle.gather(1, src[:, None])[:, 0] > 0)
    # the policy's decision stays on the GPU: one action slot per replica = (from_cluster | -1, idle position, target node)
    actions = torch.stack([torch.where(move, src, torch.full_like(src, -1)).int(), torch.zeros_like(src).int(),
                           some_node_of[dst]], dim=1).reshape(R, 1, 3).contiguous()
    env.apply_dispatch_torch(actions)                       # vds_apply_dispatch_device: asynchronous, no host round trip
    env.advance()
env.sync()
cn = env.counters()
dt = time.time() - t0
print("replicas %d, %d ticks in %.2f s (%.3g env-steps*replicas/s incl. the torch policy); rejects per replica: mean %.1f (min %d, max %d); dispatches %d" % (
    R, env.T, dt, R * env.T / dt, cn[:, 1].mean(), cn[:, 1].min(), cn[:, 1].max(), int(cn[:, 4].sum())))
