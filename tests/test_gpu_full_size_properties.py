"""BASELINE.json configs[1] and configs[3] at FULL bench size (1024 replicas x 192 clusters x 10 000 vehicles x 200 000
orders) through size-independent properties, every tick, every replica:

  * conservation: each vehicle is in exactly one container - sum(idle) + sum(in flight) == V;
  * bookkeeping: OrderNum is the same deterministic cursor for every replica; rejects + matched == OrderNum;
    the idle vehicles that disappear in a slot are exactly the orders matched in it;
  * determinism / replication: replicas that start from the same vehicle placement stay bit-identical (per-order
    vehicle and wait), wherever they sit in the batch (different workgroups, rows, wavefront groups);
  * anchoring: a few replicas against the CPU oracle (bit-exact), and the day repeated after vds_reset_again.
"""
import hashlib

import numpy as np
import pytest

from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads

pytestmark = pytest.mark.gpu

R = 1024
TWINS = [(0, 1023), (5, 514), (17, 700)]        # (source, copy): same start nodes at distant batch positions


@pytest.mark.parametrize("name", ["cfg2", "cfg4"])
def test_full_size_invariants(name):
    w = workloads.didi_day("cfg2") if name == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
    V = w.vehicles
    init = w.vehicle_nodes(R)
    for a, b in TWINS:
        init[b] = init[a]
    env = w.make_env(R)
    env.reset(init)
    T = env.T
    prev_matched = np.zeros(R, dtype=np.int64)
    for t in range(T):
        env.step()
        ob, cn = env.obs(), env.counters()
        idle, fly = ob["idle_now"].sum(axis=1), ob["inflight"].sum(axis=1)
        assert ((idle + fly) == V).all(), "tick %d: a vehicle is in no container or in two" % t
        assert (cn[:, 0] == cn[0, 0]).all(), "tick %d: OrderNum differs between replicas" % t
        assert (cn[:, 1] + cn[:, 2] == cn[:, 0]).all(), "tick %d: rejects + matched != orders" % t
        assert (ob["cl_orders"].sum(axis=1) == ob["cl_orders"][0].sum()).all()
        taken = ob["idle_pre"].sum(axis=1) - idle
        np.testing.assert_array_equal(taken, cn[:, 2] - prev_matched, err_msg="tick %d: vehicles taken != orders matched" % t)
        prev_matched = cn[:, 2].copy()
        env.advance()
    cn = env.counters()
    assert cn[0, 0] == w.release_min.size - 1                # quirk Q1: the last order is never processed
    assert (cn[:, 3] >= 0).all() and (cn[:, 7] > 0).all()
    # replicas with the same start nodes: bit-identical per-order results
    for a, b in TWINS:
        oa, ob_ = env.orders(a, 1), env.orders(b, 1)
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(oa[k][0], ob_[k][0], err_msg="replicas %d / %d differ in %s" % (a, b, k))
        assert (cn[a] == cn[b]).all()
    # anchors against the oracle
    o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server,
               w.release_min, w.pickup, w.delivery, V)
    for r in (0, 511, 1023):
        o.reset(init[r]); o.run_day()
        exp, oc, got = o.orders(), o.counters(), env.orders(r, 1)
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][0], exp[k], err_msg="replica %d %s" % (r, k))
        assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
    # the same day again from the resident start nodes: identical totals and results
    tot = env.total_counters()
    h = hashlib.sha256(env.orders(300, 2)["vehicle"].tobytes()).hexdigest()
    env.reset_again()
    env.run(T)
    assert (env.total_counters() == tot).all()
    assert hashlib.sha256(env.orders(300, 2)["vehicle"].tobytes()).hexdigest() == h
    env.close()
