"""Hooked day graph (vds_run_hooked, fixed action tensor, K = 8 with two moves per city and slot): slot time by replica-group count
and observation planes, next to the call-by-call loop.   python profiles/r05/hooked_groups.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R, K = 1024, 8
w = workloads.didi_day("cfg2")
stream = torch.cuda.current_stream()
env = w.make_env(R, stream=stream.cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
n2c = np.asarray(w.city.node2cluster)
node_of = torch.tensor([int(np.flatnonzero(n2c == c)[0]) for c in range(env.C)], dtype=torch.int32, device="cuda")
ar = torch.arange(R, device="cuda", dtype=torch.int32)
acts = torch.full((R, K, 3), -1, dtype=torch.int32, device="cuda"); acts[:, :, 1] = 0; acts[:, :, 2] = node_of[0]
acts[:, 0, 0] = ar % env.C; acts[:, 0, 2] = node_of[((ar + 97) % env.C).long()]
acts[:, 1, 0] = (ar + 97) % env.C; acts[:, 1, 2] = node_of[(ar % env.C).long()]
def sync():
    torch.cuda.synchronize()
    try: env.sync()
    except Exception as e:
        if "skipped" not in str(e): raise
def timed(f, days=3):
    f(); sync()
    t0 = time.perf_counter()
    for _ in range(days): f()
    sync()
    return (time.perf_counter() - t0) / days / T * 1e6
def eager(planes):
    env.reset_again()
    for t in range(T):
        env.step()
        if planes: env.obs_device_ptr(planes)
        env.apply_dispatch_torch(acts); env.advance()
print("hook-less vds_run: %.1f us per slot" % timed(lambda: (env.reset_again(), env.run(T))))
for planes, label in ((15, "idle_pre+idle_now+supply+orders"), (11, "without supply"), (0, "no observation planes")):
    print("%-34s call by call %.1f us per slot" % (label, timed(lambda: eager(planes))))
    for G in (1, 2, 3):
        env.set_run_groups(G, -1)
        kw = dict(idle_pre=bool(planes & 1), idle_now=bool(planes & 2), supply=bool(planes & 4), cl_orders=bool(planes & 8), inflight=False)
        print("%-34s day graph, %d group(s): %.1f us per slot" % (label, G, timed(lambda: (env.reset_again(), env.run_hooked(T, actions=acts, **kw)))))
    env.set_run_groups(0, -1)
env.close()
