"""A/B timing of two builds of libvds on ONE box (box-to-box spread on the pool is +-5 %, larger than most kernel
changes): alternates whole simulated days between the builds in separate processes, >= 3 s timed per arm and round.

    python profiles/ab.py libvds_r01.so libvds.so [--workload cfg2] [--replicas 1024] [--rounds 3] [--days N]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys, time, json
sys.path.insert(0, %r)
import torch
from vehicles_dispatch_simulator_amd import workloads
name, R, days, distinct = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = workloads.didi_day("cfg2") if name == "cfg2" else (workloads.didi_day("cfg4", neighbor=True, service_m=2000.0) if name == "cfg4" else workloads.stress())
env = w.make_env(R, load=(distinct <= 1)) if distinct <= 1 else w.make_env(R, load=False)
if distinct > 1:
    env.load_order_days(workloads.distinct_days(w, distinct), list(range(R)) if distinct >= R else [r %% distinct for r in range(R)])
env.reset(w.vehicle_nodes(R))
T = env.T
for _ in range(2):
    env.reset_again(); env.run(T)
env.sync()
t0 = time.perf_counter()
for _ in range(days):
    env.reset_again(); env.run(T)
env.sync()
dt = time.perf_counter() - t0
print(json.dumps({"ms_per_day": dt / days * 1e3, "env_steps_per_s": T * R * days / dt, "kernel": env.main_kernel()}))
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--days", type=int, default=250)
    ap.add_argument("--distinct", type=int, default=1, help="number of distinct order days (per-replica days when > 1)")
    a = ap.parse_args()
    res = {l: [] for l in a.libs}
    for _ in range(a.rounds):
        for lib in a.libs:
            # "path@NAME=VALUE[@NAME=VALUE..]": the same library under other environment switches
            parts = lib.split("@")
            env = dict(os.environ, VDS_LIB=parts[0])
            for kv in parts[1:]:
                k, v = kv.split("=", 1)
                env[k] = v
            out = subprocess.run([sys.executable, "-c", WORKER, a.workload, str(a.replicas), str(a.days), str(a.distinct)], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if out.returncode:
                print(lib, "FAILED", out.stderr.decode()[-500:])
                continue
            res[lib].append(json.loads(out.stdout.decode().strip().splitlines()[-1]))
    for lib, rs in res.items():
        if rs:
            ms = [r["ms_per_day"] for r in rs]
            print("%-24s %s  ms/day: %s  best %.3f  mean %.3f  -> %.3g env-steps*replicas/s" % (
                lib, rs[0]["kernel"], " ".join("%.3f" % m for m in ms), min(ms), sum(ms) / len(ms), rs[0]["env_steps_per_s"] * rs[0]["ms_per_day"] / min(ms)))


if __name__ == "__main__":
    main()
