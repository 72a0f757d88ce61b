"""Same-box A/B of one library switch: alternating processes, N rounds.   python profiles/ab_env.py VAR=a VAR=b [workload] [days] [rounds]
Each arm: `days` hook-less days of 1024 replicas (vds_reset_again + vds_run), wall clock around them after a warm-up."""
import os, subprocess, sys
a, b = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
days = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
CODE = r'''
import sys, time
sys.path.insert(0, ".")
import torch
from vehicles_dispatch_simulator_amd import workloads
wl, days = sys.argv[1], int(sys.argv[2])
w = workloads.didi_day("cfg2") if wl == "cfg2" else (workloads.didi_day("cfg4", neighbor=True, service_m=2000.0) if wl == "cfg4" else workloads.stress())
R = 128 if wl == "cfg5" else 1024
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
for _ in range(5): env.reset_again(); env.run(env.T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(days): env.reset_again(); env.run(env.T)
torch.cuda.synchronize()
print("%.4f" % ((time.perf_counter() - t0) / days * 1e3))
'''
res = {a: [], b: []}
for _ in range(rounds):
    for arm in (a, b):
        k, v = arm.split("=", 1)
        env = dict(os.environ); env[k] = v
        out = subprocess.run([sys.executable, "-c", CODE, wl, str(days)], env=env, capture_output=True, text=True)
        res[arm].append(float(out.stdout.strip().splitlines()[-1]))
for arm in (a, b):
    print("%-28s ms per day: %s  mean %.4f" % (arm, " ".join("%.4f" % x for x in res[arm]), sum(res[arm]) / len(res[arm])))
