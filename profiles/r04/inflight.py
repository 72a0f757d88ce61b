"""How the two chains of a grouped day really overlap ("show the overlap without the tracer").  Instrumented build:
every k_tick_dense launch stamps its first wavefront in / last wavefront out on the device-wide 100 MHz real-time counter
(s_memrealtime), per slot and
per chain; one day of configs[1] with the default vds_run (two chains), then one with one chain.
    VDS_LIB=$PWD/build/libvds_prof.so python profiles/r04/inflight.py [replicas]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = workloads.didi_day("cfg2")
env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
print(env.main_kernel(), env._lib.vds_build_id().decode(), "replicas", R)
for _ in range(3):          # (the handle settles on its per-slot forms: vds_api.hip adapt_dense)
    env.reset_again(); env.run(T); env.sync()
for groups in (0, 1):
    env.set_run_groups(groups, -1 if groups == 0 else 0)
    G = env.run_groups()
    env.reset_again(); env.run(T); env.sync()                 # graph built, warm
    env._lib.vds_debug_ablate(env._h, 1048576)
    env.reset_again(); env.sync()
    env._lib.vds_debug_read_span(env._h, None, 1)
    env.run(T); env.sync()
    buf = np.zeros(2 * 256 * 2, dtype=np.uint64)
    env._lib.vds_debug_read_span(env._h, buf.ctypes.data, 0)
    env._lib.vds_debug_ablate(env._h, 0)
    sp = buf.reshape(2, 256, 2)[:G, :T].astype(np.float64) / 100.0      # us
    t0 = sp[:, :, 0].min()
    sp -= t0
    dur = sp[:, :, 1] - sp[:, :, 0]
    day = sp[:, :, 1].max()
    print("\n%d chain(s): day %.1f us = %.2f us per slot; launch duration (first wavefront in -> last out) mean %.1f us, min %.1f, max %.1f" % (G, day, day / T, dur.mean(), dur.min(), dur.max()))
    for g in range(G):
        gap = sp[g, 1:, 0] - sp[g, :-1, 1]
        print("  chain %d: %d launches, mean duration %.1f us, gap to the chain's next launch mean %.2f us (min %.2f, max %.2f)" % (g, T, dur[g].mean(), gap.mean(), gap.min(), gap.max()))
    # launches in flight over time (event sweep)
    ev = sorted([(x, 1) for x in sp[:, :, 0].ravel()] + [(x, -1) for x in sp[:, :, 1].ravel()])
    infl, last, hist = 0, 0.0, {}
    for x, d in ev:
        hist[infl] = hist.get(infl, 0.0) + (x - last)
        infl += d; last = x
    print("  launches in flight: " + ", ".join("%d: %.1f %%" % (k, 100 * v / day) for k, v in sorted(hist.items())))
    if G == 2:
        off = sp[1, :, 0] - sp[0, :, 0]
        print("  chain 1 starts %.1f us after chain 0 on average (slot by slot: min %.1f, max %.1f); sum of the two chains' launch durations %.1f us vs day %.1f us" % (off.mean(), off.min(), off.max(), dur.sum(), day))
