"""Life cycle of vds_run's day graph with replica groups (parallel branches), repeated: build / launch, then (flow >= 1) a
second and third graph of other shapes on the same handle (the previous one is dropped each time), destroy.  Used to pin down
a heap corruption inside the HIP runtime (ROCm 7.0 libamdhip64 bundled with torch 2.10) when such a graph is destroyed:
notes.md in this directory.

    python profiles/r03_run_groups/crash_probe.py <groups> <stagger> <flow 0|1|2|3> <repetitions>
"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_golden
from test_gpu_replica_days import mk_env, synth_days
from vehicles_dispatch_simulator_amd import synth
groups, stagger, flow, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = load_golden("tiny_kmeans_dfs2")
R = 40
day = synth_days(g, 1, seed=77)
days5 = synth_days(g, 5, seed=901)
valid = g["node2cluster"] >= 0
init = np.stack([synth.init_vehicle_nodes(random.Random(400 + r), int(g["N"]), int(g["V"]), valid) for r in range(R)]).astype(np.int32)
for rep in range(reps):
    env = mk_env(g, R)
    env.load_orders(*day[0])
    env.set_run_groups(groups, stagger)
    env.reset(init)
    env.run(env.T); env.sync()
    a = env.counters().copy()
    if flow >= 1:
        env.reset_again()
        k = env.T // 3
        env.run(k)
        if flow >= 2:
            env.step(); env.advance()
            env.run(env.T - k - 1)
        else:
            env.run(env.T - k)
        env.sync()
        assert np.array_equal(a, env.counters())
    if flow >= 3:
        # new order tables on the same handle, several times: the day graph is updated in place (hipGraphExecUpdate) when the day
        # has as many slots as the last one, re-built otherwise; every day against a one-group handle
        for d in (0, 1, 3, 0):
            env.load_orders(*days5[d]); env.reset(init); env.run(env.T); env.sync()
            ref = mk_env(g, R); ref.load_orders(*days5[d]); ref.set_run_groups(1, 0); ref.reset(init); ref.run(ref.T); ref.sync()
            assert np.array_equal(ref.counters(), env.counters()), d
            ref.close()
    env.close()
print("ok")
