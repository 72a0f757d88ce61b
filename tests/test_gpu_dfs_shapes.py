"""Neighbour-search mode on shapes the golden fixtures do not reach, each replica against the CPU oracle (bit-exact),
through all neighbour-search engines (default: hybrid tick = k_tick_rows in stamp mode + k_dfs_walk; k_tick_replica2 =
force_generic 3: lower-bound rounds, what the hybrid tick falls back to; force_generic 1: the serial form k_match_dfs):
visit sequences longer than 256 clusters (several batches per wavefront), idle lists far beyond the register tables,
more than 16 arrivals of one bucket in one tick, every vehicle starting in a handful of clusters."""
import random

import numpy as np
import pytest

from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv, synth
from vehicles_dispatch_simulator_amd.env import neighbors_to_csr

pytestmark = pytest.mark.gpu


def check(city, depth, V, O, oseed, R, init, pick=None, dele=None, kernel0="k_dfs_dense", **kw):
    start, p0, d0 = synth.make_orders(oseed, city.N, O)
    pick = p0 if pick is None else pick
    dele = d0 if dele is None else dele
    rel = synth.release_minutes(start)
    off, idx = neighbors_to_csr(city.neighbors)
    results = {}
    import os
    # mode 8: the hybrid tick on the WIDE layout (VDS_DENSE_DFS=0, read when the static tables are loaded: k_tick_rows in stamp mode + the
    # committing walk) - what the library keeps for order days per replica, costs beyond a byte, orders without a static arrival slot
    switches = {8: {"VDS_DENSE_DFS": "0"}}
    extra = (8,) if kernel0 == "k_dfs_dense" else ()
    for mode in (0, 3, 1) + extra:
        os.environ.update(switches.get(mode, {}))
        try:
            env = BatchedDispatchEnv(city.cost, city.node2cluster, off, idx, replicas=R, vehicles=V, depth_limit=depth,
                                     neighbor_can_server=True, force_generic=0 if mode == 8 else mode, idle_cap=min(1024, max(64, V)), ring_cap=max(64, V), far_cap=max(64, V), **kw)
            env.load_orders(rel, pick, dele)
        finally:
            for k in switches.get(mode, {}):
                os.environ.pop(k, None)
        assert env.main_kernel() == {0: kernel0, 3: "k_tick_replica2", 1: "k_match_dfs", 8: "k_dfs_hybrid"}[mode]
        env.reset(init)
        env.run(env.T)
        results[mode] = (env.orders(), env.counters(), env.obs(), [env.lists(r) for r in range(R)])
        env.close()
    for r in range(R):
        o = Oracle(city.cost, city.node2cluster, off, idx, depth, True, rel, pick, dele, V)
        o.reset(init[r])
        o.run_day()
        exp, oc, ol = o.orders(), o.counters(), o.lists()
        for mode, (got, cn, obs, lists) in results.items():
            for k in ("status", "vehicle", "wait"):
                np.testing.assert_array_equal(got[k][r], exp[k], err_msg="mode %d replica %d %s" % (mode, r, k))
            for i, k in enumerate(("order_num", "reject_num", "matched", "wait_sum")):
                assert cn[r, i] == oc[k], (mode, r, k)
            assert cn[r, 6] == oc["sum_order_value"] and cn[r, 7] == oc["evals"], (mode, r)
            np.testing.assert_array_equal(obs["idle_now"][r], o.obs()["idle_now"])
            np.testing.assert_array_equal(lists[r]["idle_off"], ol["idle_off"])
            np.testing.assert_array_equal(lists[r]["idle_veh"][:ol["idle_off"][-1]], ol["idle_veh"][:ol["idle_off"][-1]])
    return results[0][1]


def test_long_visit_sequences_many_clusters():
    city = synth.make_city(seed=901, N=1500, C=400)
    seqs = synth.dfs_sequences(city.neighbors, 5)
    assert max(len(s) for s in seqs) > 4 * 64            # more than one 64-cluster batch for some wavefront
    V, R = 500, 3
    init = np.stack([synth.init_vehicle_nodes(random.Random(31 + r), city.N, V) for r in range(R)])
    cn = check(city, 5, V, 9000, 77, R, init, kernel0="k_tick_replica2")      # beyond the hybrid tick's 256 visited clusters
    assert cn[:, 1].min() >= 0


def test_visit_sequences_of_several_batches_hybrid():
    city = synth.make_city(seed=905, N=1500, C=260)
    seqs = synth.dfs_sequences(city.neighbors, 3)
    assert 2 * 64 < max(len(s) for s in seqs) <= 256     # three or four 64-cluster batches per dry order, still the hybrid tick
    V, R = 500, 3
    init = np.stack([synth.init_vehicle_nodes(random.Random(41 + r), city.N, V) for r in range(R)])
    check(city, 3, V, 9000, 83, R, init)


def test_all_vehicles_start_in_a_few_clusters_long_lists():
    city = synth.make_city(seed=902, N=900, C=60)
    V, R = 1000, 2
    nodes = np.flatnonzero(city.node2cluster < 4)
    init = np.stack([nodes[synth.uniform_int(50 + r, 3, np.arange(V), nodes.size)] for r in range(R)]).astype(np.int32)
    check(city, 2, V, 12000, 78, R, init)


def test_everything_is_delivered_to_one_cluster_bursts_of_arrivals():
    city = synth.make_city(seed=903, N=800, C=48)
    V, R, O = 900, 2, 14000
    start, pick, dele = synth.make_orders(79, city.N, O)
    target = np.flatnonzero(city.node2cluster == 7)
    dele = target[synth.uniform_int(5, 9, np.arange(O), target.size)].astype(np.int32)
    init = np.stack([synth.init_vehicle_nodes(random.Random(61 + r), city.N, V) for r in range(R)])
    cn = check(city, 2, V, O, 79, R, init, pick=pick, dele=dele)
    assert (cn[:, 2] > 800).all()                          # plenty of matches: vehicles pile up in cluster 7 and are found by the search


def test_hundreds_of_dry_orders_per_slot():
    """Few vehicles, many orders, every pickup in a handful of clusters: hundreds of dry orders per slot and replica - more than one
    64-entry chunk of the hybrid tick's steal log, its record pool going round many times, redo chains that empty buckets."""
    city = synth.make_city(seed=906, N=700, C=36)
    V, R, O = 3000, 3, 40000
    start, pick, dele = synth.make_orders(85, city.N, O)
    hot = np.flatnonzero(city.node2cluster < 6)
    pick = hot[synth.uniform_int(6, 11, np.arange(O), hot.size)].astype(np.int32)
    init = np.stack([synth.init_vehicle_nodes(random.Random(71 + r), city.N, V) for r in range(R)])
    cn = check(city, 2, V, O, 85, R, init, pick=pick, dele=dele)
    assert (cn[:, 2] > 15000).all()      # on average more than 100 served per slot, most of them by a neighbour (30 of 36 clusters see no pickup)
