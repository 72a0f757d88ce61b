"""TEST INFRASTRUCTURE - one worker of bench.py's all-cores `cpu_baseline_all_cores` leg.

Runs the CPU oracle (oracle/vds_oracle.c, the C restatement of the reference's per-slot loop,
simulator.py:1036-1092) on whole single-replica days of a workload dumped by bench.py, for a fixed
time budget, and prints one line "days ticks evals seconds".  Never imported by the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    path, budget, index = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    from oracle.oracle import Oracle
    z = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r") for k in
         ("cost", "node2cluster", "nbr_off", "nbr_idx", "release_min", "pickup", "delivery", "init", "scalars")}
    depth, ncs, V = (int(x) for x in z["scalars"])
    o = Oracle(np.asarray(z["cost"]), np.asarray(z["node2cluster"]), np.asarray(z["nbr_off"]), np.asarray(z["nbr_idx"]), depth, bool(ncs),
               np.asarray(z["release_min"]), np.asarray(z["pickup"]), np.asarray(z["delivery"]), V)
    init = np.asarray(z["init"])
    days = ticks = evals = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        o.reset(init[(index + days) % len(init)])
        ticks += o.run_day()
        evals += o.counters()["evals"]
        days += 1
    print(days, ticks, evals, time.perf_counter() - t0)


if __name__ == "__main__":
    main()
