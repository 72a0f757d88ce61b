"""Determinism soak (python profiles/soak.py [days] [cfg2|cfg4]): the same 1024-replica day repeated N times must give bit-identical totals and per-order
results every time (atomics only allocate slots; the result never depends on their order)."""
import sys, hashlib
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
w = workloads.didi_day("cfg2") if wl == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
R = 1024
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
ref_tot, ref_hash = None, None
for d in range(N):
    env.reset_again(); env.run(env.T)
    tot = env.total_counters()
    o = env.orders(1000, 2)
    h = hashlib.sha256(o["vehicle"].tobytes() + o["wait"].tobytes()).hexdigest()
    if ref_tot is None:
        ref_tot, ref_hash = tot, h
    assert (tot == ref_tot).all() and h == ref_hash, (d, tot, ref_tot)
print("soak ok (%s): %d identical days, totals %s" % (wl, N, ref_tot.tolist()))
