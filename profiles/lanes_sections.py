"""In-kernel section attribution of k_tick_lanes' fast path (instrumented build: make -C vehicles_dispatch_simulator_amd/csrc prof).

    VDS_LIB=libvds_prof.so python profiles/lanes_sections.py [lg=..,loc=..,keys=..]
Prints, per lanes-per-bucket class, the mean cycles a wavefront spends between the stamps (each stamp waits for everything in
flight, so a section owns the latency of what it issued)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for spec in sys.argv[1:2]:
    for kv in spec.split(","):
        k, v = kv.split("=")
        os.environ["VDS_LANES_" + k.upper()] = v
from vehicles_dispatch_simulator_amd import workloads

NAMES = ["0 block / bucket descriptors (scalar loads)", "1 header + prefetched list / ring words + cost block", "2 list -> LDS (further batches)", "3 arrivals ranked + appended",
         "4 order records of a chunk", "5 match loop", "6 results: out + atomics issued", "7 removal in LDS", "8 list tail back to HBM", "9 entry stores, header, counters"]
w = workloads.didi_day("cfg2")
R = 1024
env = w.make_env(R, force_generic=6)
env.reset(w.vehicle_nodes(R))
env.run(env.T)
env.sync()
buf = np.zeros(64, dtype=np.uint64)
env._lib.vds_debug_lanes_prof(env._h, buf.ctypes.data_as(C.c_void_p))     # clear
env.reset_again(); env.run(env.T); env.sync()
env._lib.vds_debug_lanes_prof(env._h, buf.ctypes.data_as(C.c_void_p))
buf = buf.reshape(4, 16)
for lg in range(4):
    n = int(buf[lg, 15])
    if not n:
        continue
    tot = float(buf[lg, :10].sum())
    print("L = %d: %d fast wavefronts with orders in the day, %.0f cycles each; %d order-less; %d on the slow path, %.0f cycles each" % (
        1 << lg, n, tot / n, int(buf[lg, 12]), int(buf[lg, 13]), buf[lg, 11] / max(1, int(buf[lg, 13]))))
    for i, nm in enumerate(NAMES):
        print("   %-58s %8.0f  %5.1f %%" % (nm, buf[lg, i] / n, 100.0 * buf[lg, i] / tot))
env.close()
