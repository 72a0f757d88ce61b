"""Timing-only ablation of ONE tick of k_tick_lanes from an identical state: the day runs normally up to tick t0, then that
tick is launched with Static.lane_ablate = f (vds_debug_lanes_ablate) and timed with HIP events; reset, repeat.

    python profiles/lanes_tick_ablate.py [lg=2,loc=32,keys=8] [t0 ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = sys.argv[1] if len(sys.argv) > 1 else "lg=2,loc=32,keys=8"
for kv in spec.split(","):
    k, v = kv.split("=")
    os.environ["VDS_LANES_" + k.upper()] = v
from vehicles_dispatch_simulator_amd import workloads

ticks = [int(x) for x in sys.argv[2:]] or [60, 110]
w = workloads.didi_day("cfg2")
R = 1024
stream = torch.cuda.current_stream()
env = w.make_env(R, force_generic=6, stream=stream.cuda_stream)
env.reset(w.vehicle_nodes(R))
FLAGS = [(0, "full"), (1, "no counter atomics"), (2, "no result stores"), (4, "no arrival posts (atomic + entry)"), (8, "no list write-back"), (16, "no header store"), (31, "none of them")]
for t0 in ticks:
    print("tick %d" % t0)
    for f, name in FLAGS:
        ts = []
        for rep in range(4):
            env.reset_again()
            env._lib.vds_debug_lanes_ablate(env._h, 0)
            for _ in range(t0):
                env.step(); env.advance()
            env._lib.vds_debug_lanes_ablate(env._h, f)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); env.step(); b.record(stream); b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        print("   %-36s %7.1f us" % (name, min(ts)))
env._lib.vds_debug_lanes_ablate(env._h, 0)
env.close()
