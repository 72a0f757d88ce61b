"""Do replica groups overlap when each is its own handle on its own stream?  configs[3], G handles of 1024/G replicas."""
import sys, time
sys.path.insert(0, ".")
import torch
from vehicles_dispatch_simulator_amd import workloads
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
def run(G, days=6, R=None):
    R = R or 1024 // G // 32 * 32
    streams = [torch.cuda.Stream() for _ in range(G)]
    envs = []
    for s in streams:
        e = w.make_env(R, stream=s.cuda_stream)
        e.set_run_groups(1) if hasattr(e, "set_run_groups") else None
        e.reset(w.vehicle_nodes(R)); envs.append(e)
    for _ in range(2):
        for e in envs: e.reset_again(); e.run(e.T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(days):
        for e in envs: e.reset_again(); e.run(e.T)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / days * 1e3
    print("G=%d handles x %d replicas: %.2f ms per day -> %.2f M" % (G, R, ms, G * R * envs[0].T / ms / 1e3), flush=True)
import os
if os.environ.get('ALONE'):
    for R in [int(x) for x in os.environ["ALONE"].split(",")]:
        run(1, R=R)
else:
    for G in (1, 2, 3, 4):
        run(G)
