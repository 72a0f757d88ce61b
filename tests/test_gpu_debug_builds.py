"""The instrumented builds of libvds run through parity workloads in subprocesses (``VDS_LIB`` selects the library):

* ``make dbg`` (``-DWKDEBUG``): k_dfs_walk checks itself - bounds of every winner, evaluations of the dry orders counted a
  second way - and prints ``k_dfs_walk check N failed`` when something is off;
* ``make canary`` (``-DVDS_CANARY``): every device table sits between poisoned guard zones; an out-of-bounds write damages a
  guard (``vds_debug_check_guards``, called by ``env.close()``), an out-of-bounds read returns the poison instead of a
  neighbour's data and shows up as a wrong result.

The libraries are built on demand into ``build/`` (about a minute each)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vehicles_dispatch_simulator_amd", "csrc")


def build(target, name):
    """The instrumented library, built here only when it is missing or was built from other sources: every library carries the content
    hash of its sources (``vds_build_id``: ``src:<kernels>+<host>``, ``make srchash``), which is compared - not the files' times, which a
    copy of the tree to another machine does not keep in order."""
    path = os.path.join(ROOT, "build", name)
    want = subprocess.check_output(["make", "-s", "-C", CSRC, "srchash"], text=True).strip().splitlines()[-1].encode()
    if not os.path.exists(path) or want not in open(path, "rb").read():
        subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, target])
    return path


WORKER = r"""
import sys, time
T0 = time.perf_counter()
import numpy as np
sys.path.insert(0, %r)
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads, _lib
assert _lib.load().vds_build_id().decode().endswith(%r), _lib.load().vds_build_id()
for neighbor, fg, R, veh, dd in %r:
    t1 = time.perf_counter()
    w = workloads.tiny(neighbor=neighbor, vehicles=veh, orders=4000)
    init = w.vehicle_nodes(R)
    env = w.make_env(R, force_generic=fg, dense_debug=dd)
    env.reset(init)
    env.run(env.T)
    got, cn = env.orders(), env.counters()
    for r in range(0, R, max(1, R // 6)):
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, w.release_min, w.pickup, w.delivery, w.vehicles)
        o.reset(init[r]); o.run_day()
        exp, oc = o.orders(), o.counters()
        for k in ("status", "vehicle", "wait"):
            assert np.array_equal(got[k][r], exp[k]), (neighbor, fg, r, k)
        assert cn[r, 7] == oc["evals"] and cn[r, 1] == oc["reject_num"]
    t2 = time.perf_counter()
    print("ok", neighbor, fg, env.main_kernel())
    env.close()          # guarded build: raises when a guard zone was written
    print("   %%.2f s engine + oracle, %%.2f s close, %%.2f s since start" %% (t2 - t1, time.perf_counter() - t2, time.perf_counter() - T0))
print("WORKER DONE")
"""


def run_worker(lib, suffix, cases, **environ):
    env = dict(os.environ, VDS_LIB=lib, **environ)
    p = subprocess.run([sys.executable, "-c", WORKER % (ROOT, suffix, cases)], env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "WORKER DONE" in out, out[-3000:]
    return out


def test_self_checking_walk_build():
    lib = build("dbg", "libvds_dbg.so")
    # scarce vehicles: most orders run dry, redo chains and rejects; two replica counts
    out = run_worker(lib, "+dbg", [(True, 0, 8, 40, None), (True, 0, 37, 25, None), (True, 0, 5, 90, None)])
    assert "k_dfs_dense" in out
    assert "check" not in out.replace("vds_debug_check", ""), out[-3000:]
    # the same on the wide layout (k_tick_rows in stamp mode + the committing walk)
    out = run_worker(lib, "+dbg", [(True, 0, 8, 40, None), (True, 0, 5, 90, None)], VDS_DENSE_DFS="0")
    assert "k_dfs_hybrid" in out and "check" not in out.replace("vds_debug_check", ""), out[-3000:]


def test_guarded_build_all_tick_paths():
    lib = build("canary", "libvds_canary.so")
    cases = [(False, 0, 19, 150, None), (False, 0, 37, 150, (8, 0, 0, 0)), (False, 0, 37, 150, (4, 0, 0, 0)), (False, 0, 21, 150, (16, 12, 2, 0)), (False, 0, 5, 150, (16, 0, 0, 1)),
             (False, 1, 5, 150, None), (False, 5, 37, 150, None), (True, 0, 9, 40, None), (True, 3, 6, 40, None), (True, 1, 3, 40, None)]
    out = run_worker(lib, "+canary", cases)
    assert out.count("ok ") == len(cases)
    # neighbour search with scarce vehicles - most orders dry - on both layouts
    da = [(True, 0, 9, 40, None), (True, 0, 37, 25, None), (True, 0, 5, 90, None)]
    out = run_worker(lib, "+canary", da)
    assert out.count("ok ") == len(da) and out.count("k_dfs_dense") == len(da)
    out = run_worker(lib, "+canary", da[:2], VDS_DENSE_DFS="0")
    assert out.count("ok ") == 2 and out.count("k_dfs_hybrid") == 2


def test_guarded_build_replica_days_and_dispatch():
    """The tests that move the most tables (order days per replica, the map changes, dispatch, the list image of the reset, the hooked day graph, observations in place) under the guarded library."""
    lib = build("canary", "libvds_canary.so")
    env = dict(os.environ, VDS_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_replica_days.py"),
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_edge_cases.py"), os.path.join(ROOT, "tests", "test_gpu_run_hooked.py"), "-k",
                        "(every_replica_replays and (fast or dense16 or rows)) or changes_every_episode or blocks_of_sixteen or tiny_dispatch-fast or tiny_dispatch-dense16 or tiny_dispatch_dfs2-fast or device_resident or (burst and dense16)"
                        " or reset_again_of_a_city or observations_in_place or (hooked_day_graph and tiny_kmeans-1)"],
                       env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]


def test_guard_check_finds_a_damaged_guard():
    lib = build("canary", "libvds_canary.so")
    code = r"""
import sys
sys.path.insert(0, %r)
from vehicles_dispatch_simulator_amd import workloads
w = workloads.tiny()
env = w.make_env(3)
env.reset(w.vehicle_nodes(3))
env.run(5)
assert env._lib.vds_debug_check_guards(env._h) == 0
assert env._lib.vds_debug_poke_guard(env._h) == 0
try:
    env.close()
except Exception as e:
    assert "guard zones damaged" in str(e), e
    print("GUARD FOUND")
""" % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VDS_LIB=lib), capture_output=True, text=True, timeout=600)
    assert "GUARD FOUND" in p.stdout, (p.stdout + p.stderr)[-2000:]
