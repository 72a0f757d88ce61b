"""k_tick_dense picks its form per SLOT (``vds_api.hip`` ``adapt_dense``): 8 lanes per replica with 128-entry tables where the lists are
short, 16 lanes with 256-entry tables where many buckets would leave the fast path.  Results must not depend on it.  In subprocesses
(the switches are environment variables read when a day is loaded):
  * the two forms alternating slot by slot at 96 replicas (32-row workgroups <-> 16-row workgroups), with and without neighbour search;
  * the adaptive path itself: a city whose lists outgrow the 128-entry tables in some slots, a low threshold - after three episodes some
    slots (not all) run the second form, the day graph was rebuilt, and every checked replica still equals the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, %r)
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads
mode = %r
for neighbor, veh, R, episodes, lim in %r:
    if lim is not None:
        os.environ["VDS_DENSE_TICK_LIM"] = str(lim)       # (read when the day is loaded)
    w = workloads.tiny(neighbor=neighbor, vehicles=veh, orders=4000)
    init = w.vehicle_nodes(R)
    env = w.make_env(R)
    env.reset(init)
    for ep in range(episodes):
        if ep:
            env.reset_again()
        env.run(env.T)
        env.sync()
    forms = np.zeros(env.T, dtype=np.uint8); n = C.c_int32()
    assert env._lib.vds_debug_tick_forms(env._h, forms.ctypes.data_as(C.c_void_p), env.T, C.byref(n)) == 0
    n16 = int(forms[:n.value].sum())
    if mode == "alt":
        assert n.value == env.T and n16 == env.T // 2, (n.value, n16)
    else:
        assert 0 < n16 < env.T, "expected a mixed day: %%d of %%d slots in the second form" %% (n16, env.T)
    got, cn = env.orders(), env.counters()
    for r in range(0, R, max(1, R // 8)):
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, w.release_min, w.pickup, w.delivery, w.vehicles)
        o.reset(init[r]); o.run_day()
        exp, oc = o.orders(), o.counters()
        for k in ("status", "vehicle", "wait"):
            assert np.array_equal(got[k][r], exp[k]), (neighbor, r, k)
        assert cn[r, 7] == oc["evals"] and cn[r, 1] == oc["reject_num"]
    print("ok", mode, neighbor, env.main_kernel(), "second form in", n16, "of", env.T, "slots; slow path buckets", env.work()["slow_path_buckets"])
    env.close()
print("WORKER DONE")
"""


def run_worker(mode, cases, **environ):
    env = dict(os.environ, **environ)
    p = subprocess.run([sys.executable, "-c", WORKER % (ROOT, mode, cases)], env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "WORKER DONE" in out, out[-3000:]
    return out


def test_forms_alternating_slot_by_slot():
    run_worker("alt", [(False, 700, 96, 1, None), (True, 700, 96, 1, None)], VDS_DENSE_TICK_FORMS="alt")


def test_per_slot_choice_adapts_and_keeps_results():
    # (limits chosen so that some slots, fewer than 85 % of them, switch: 114 / 87 of 148; nearly every slot would switch the whole day)
    out = run_worker("adapt", [(False, 1000, 96, 4, 100), (True, 1000, 96, 4, 120)])
    assert out.count("ok adapt") == 2
