"""vds_run in neighbour-search mode runs the replicas as independent GROUPS on streams (vds_set_run_groups: the stamp-mode
k_tick_rows of one group under the k_dfs_walk of the others).  Replicas never interact between hooks (simulator.py:1048-1091),
so nothing may depend on the grouping: every replica still equals its own oracle - per order, counters, container order -
for every group count / stagger mode, with and without the hipGraph, for ragged replica counts, per-replica order days
(block-wise and regrouped), partial runs and runs mixed with hooked steps."""
import os
import random

import numpy as np
import pytest

from helpers import load_golden
from test_gpu_replica_days import mk_env, mk_oracle, synth_days
from vehicles_dispatch_simulator_amd import synth

pytestmark = pytest.mark.gpu


def _init(g, R, seed):
    valid = g["node2cluster"] >= 0
    return np.stack([synth.init_vehicle_nodes(random.Random(seed + r), int(g["N"]), int(g["V"]), valid) for r in range(R)]).astype(np.int32)


def _check(env, g, days, replica_day, init, replicas=None):
    got, cn = env.orders(), env.counters()
    cache = {}
    for r in (range(env.R) if replicas is None else replicas):
        o = mk_oracle(g, days[replica_day[r]])
        o.reset(init[r]); o.run_day()
        exp, oc = o.orders(), o.counters()
        n = exp["status"].size
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][r][:n], exp[k], err_msg="replica %d %s" % (r, k))
        assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
        L, G = o.lists(), env.lists(r)
        for k in ("idle_off", "idle_veh", "arr_off", "arr_veh", "arr_min"):
            np.testing.assert_array_equal(G[k], L[k], err_msg="replica %d %s" % (r, k))


@pytest.mark.parametrize("graph", ["graph", "eager"])
@pytest.mark.parametrize("groups,stagger", [(2, 0), (2, 2), (3, 1), (3, 2), (16, 2)])
def test_shared_day_every_replica_equals_the_oracle(groups, stagger, graph, monkeypatch):
    monkeypatch.setenv("VDS_RUN_GRAPH", "1" if graph == "graph" else "0")
    g = load_golden("tiny_kmeans_dfs2")
    R = 40                                           # 3 chunks of 16, the last one ragged
    day = synth_days(g, 1, seed=77)
    init = _init(g, R, 400)
    env = mk_env(g, R)
    env.load_orders(*day[0])
    assert env.main_kernel() == "k_dfs_dense"
    env.set_run_groups(groups, stagger)
    assert env.run_groups() == (min(groups, 3) if graph == "graph" else 1)       # (groups exist only as branches of the day's graph)
    env.reset(init)
    env.run(env.T)
    env.sync()
    _check(env, g, day, np.zeros(R, dtype=np.int32), init)
    # a second day on the same handle (the graph is replayed), this time in two runs with a hooked step in between
    env.reset_again()
    k = env.T // 3
    env.run(k)
    env.step(); env.advance()
    env.run(env.T - k - 1)
    env.sync()
    _check(env, g, day, np.zeros(R, dtype=np.int32), init, replicas=[0, 15, 16, 33, 39])
    env.close()


@pytest.mark.parametrize("name", ["tiny_kmeans_dfs2", "tiny_empty_clusters_dfs2", "tiny_grid_nbr_scarce"])
@pytest.mark.parametrize("layout", ["blocks", "interleaved"])
def test_per_replica_days_in_groups(name, layout):
    g = load_golden(name)
    R = 40
    days = synth_days(g, 3, seed=333)
    replica_day = np.array([0] * 16 + [2] * 16 + [1] * 8, dtype=np.int32) if layout == "blocks" else (np.arange(R) % 3).astype(np.int32)
    init = _init(g, R, 90)
    env = mk_env(g, R)
    env.load_order_days(days, replica_day)
    if env.main_kernel() != "k_dfs_hybrid":
        env.close()
        pytest.skip("fixture does not take the hybrid tick")
    env.set_run_groups(3, 2)
    assert env.run_groups() == 3                     # (also with the replicas stored regrouped by day: interleaved map)
    env.reset(init)
    env.run(env.T)
    env.sync()
    _check(env, g, days, replica_day, init)
    env.close()


@pytest.mark.parametrize("layout", ["blocks", "interleaved"])
def test_plain_tick_per_replica_days_in_groups(layout):
    g = load_golden("tiny_kmeans")
    R = 40                                           # interleaved: 14 + 13 + 13 replicas -> stored as 3 x 16 (regrouped by day)
    days = synth_days(g, 3, seed=21)
    replica_day = np.array([0] * 16 + [2] * 16 + [1] * 8, dtype=np.int32) if layout == "blocks" else (np.arange(R) % 3).astype(np.int32)
    init = _init(g, R, 17)
    env = mk_env(g, R)
    env.load_order_days(days, replica_day)
    assert env.main_kernel() == "k_tick_dense"
    env.set_run_groups(3, 0)
    assert env.run_groups() == 3
    env.reset(init)
    env.run(env.T)
    env.sync()
    _check(env, g, days, replica_day, init)
    env.close()


def test_grouping_does_not_change_a_bit_and_is_the_default_from_512_replicas():
    """512 replicas of a tiny city (the default picks groups from 512 replicas on): the grouped default, one group and four
    free-running groups give identical per-replica counters, observations and order results."""
    g = load_golden("tiny_kmeans_dfs2")
    R = 512
    day = synth_days(g, 1, seed=5)
    init = _init(g, 8, 1)[np.arange(R) % 8]
    outs = []
    for groups, stagger in ((0, -1), (1, 0), (4, 0)):
        env = mk_env(g, R)
        env.load_orders(*day[0])
        assert env.run_groups() == 2                 # the default at 512 replicas
        env.set_run_groups(groups, stagger)
        assert env.run_groups() == (groups or 2)
        env.reset(init)
        env.run(env.T)
        env.sync()
        ob = env.obs()
        outs.append((env.counters().copy(), {k: np.array(v) for k, v in ob.items()}, {k: np.array(v) for k, v in env.orders().items()}))
        env.close()
    for cn, ob, od in outs[1:]:
        np.testing.assert_array_equal(cn, outs[0][0])
        for k in ob: np.testing.assert_array_equal(ob[k], outs[0][1][k], err_msg=k)
        for k in od: np.testing.assert_array_equal(od[k], outs[0][2][k], err_msg=k)
    np.testing.assert_array_equal(outs[0][0][:8], outs[0][0][504:512])           # same start nodes -> same day


@pytest.mark.parametrize("name,groups", [("tiny_kmeans_dfs2", 3), ("tiny_kmeans", 2), ("tiny_kmeans_dfs2", 1)])
def test_new_order_tables_on_one_handle_update_the_day_graph_in_place(name, groups):
    """vds_load_orders between days: a day with as many slots as the last one keeps the executable graph (its kernel parameters
    are replaced: hipGraphExecUpdate), another slot count re-builds it - every day equals the oracle either way; plain tick
    (two groups of k_tick_dense) and hybrid tick."""
    g = load_golden(name)
    R = 40
    base = synth_days(g, 2, seed=21)
    a = base[0]
    rng = np.random.RandomState(3)
    perm = rng.permutation(a[1].size)
    b = (a[0], a[1][perm].copy(), a[2][perm].copy())          # same release minutes (same slots), other trips
    init = _init(g, R, 700)
    env = mk_env(g, R)
    for day in (a, b, base[1], a):
        env.load_orders(*day)
        env.set_run_groups(groups, 1)
        env.reset(init)
        env.run(env.T)
        env.sync()
        _check(env, g, [day], np.zeros(R, dtype=np.int32), init, replicas=[0, 7, 16, 31, 39])
    env.close()


def test_day_graph_life_cycle_many_handles_two_alive_on_their_own_streams():
    """Branched executable graphs are never destroyed (csrc/vds_api.hip: graph pool): handles come and go - create / run / another
    day (same slots: in-place update; other slots: another shape) / run in parts / destroy - fifty times, always two handles alive
    at once on their own torch streams with their days in flight together, plain and hybrid tick.  Every day equals the
    one-group run of the same inputs; the pool stays bounded by shapes x handles alive at once (nothing is destroyed, nothing
    accumulates per handle)."""
    import torch
    from vehicles_dispatch_simulator_amd import _lib
    lib = _lib.load()
    R = 40
    cases = []
    for name, groups in (("tiny_kmeans_dfs2", 3), ("tiny_kmeans", 2)):
        g = load_golden(name)
        days = synth_days(g, 3, seed=31)
        init = _init(g, R, 900)
        refs = []
        for d in days:                               # one-group reference results of every day
            ref = mk_env(g, R); ref.load_orders(*d); ref.set_run_groups(1, 0); ref.reset(init); ref.run(ref.T); ref.sync()
            refs.append((ref.counters().copy(), {k: np.array(v) for k, v in ref.orders().items()}))
            ref.close()
        cases.append((g, groups, days, init, refs))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    pool_sizes = []
    for rep in range(25):
        envs = []
        for i, (g, groups, days, init, refs) in enumerate(cases):
            env = mk_env(g, R, stream=streams[i].cuda_stream)
            env.load_orders(*days[rep % 3]); env.set_run_groups(groups, rep % 3); env.reset(init)
            assert env.run_groups() == groups
            envs.append(env)
        for env in envs: env.run(env.T)              # both days in flight together
        for env, (g, groups, days, init, refs) in zip(envs, cases):
            env.sync()
            np.testing.assert_array_equal(env.counters(), refs[rep % 3][0])
        # another day on the same handles, run in two parts around a hooked step
        for env, (g, groups, days, init, refs) in zip(envs, cases):
            env.load_orders(*days[(rep + 1) % 3]); env.reset(init)
            k = env.T // 2
            env.run(k); env.step(); env.advance(); env.run(env.T - k - 1)
        for env, (g, groups, days, init, refs) in zip(envs, cases):
            env.sync()
            cn, od = refs[(rep + 1) % 3]
            np.testing.assert_array_equal(env.counters(), cn)
            got = env.orders()
            for key in od: np.testing.assert_array_equal(got[key], od[key], err_msg=key)
        for env in envs: env.close()
        pool_sizes.append(lib.vds_debug_graph_pool_size())
    assert max(pool_sizes) <= 48 and pool_sizes[-1] == pool_sizes[-4] == pool_sizes[-7], pool_sizes       # steady: no growth over two more 3-cycles


def test_plain_tick_groups_equal_the_oracle_and_are_the_default_from_256_replicas():
    g = load_golden("tiny_kmeans")
    day = synth_days(g, 1, seed=8)
    R = 40
    init = _init(g, R, 300)
    for groups, stagger in ((2, 0), (3, 2)):
        env = mk_env(g, R)
        env.load_orders(*day[0])
        assert env.main_kernel() == "k_tick_dense"
        env.set_run_groups(groups, stagger)
        assert env.run_groups() == groups
        env.reset(init)
        env.run(env.T)
        env.sync()
        _check(env, g, day, np.zeros(R, dtype=np.int32), init)
        env.close()
    for R_, G_ in ((512, 2), (256, 2), (255, 1)):       # (the plain tick: two chains from 256 replicas on)
        env = mk_env(g, R_)
        env.load_orders(*day[0])
        assert env.run_groups() == G_
        env.close()


@pytest.mark.parametrize("R,groups", [(70, 1), (112, 2), (112, 3), (129, 2), (200, 1)])
def test_32_row_workgroups_ragged_replica_counts_and_group_boundaries(R, groups):
    """One shared day at 8 lanes per replica from 64 replicas on: k_tick_dense takes 32 replicas per workgroup.  Replica counts
    that are no multiple of 32 (the last workgroup's tail rows do not exist) and replica groups whose boundaries are multiples of
    16 but not of 32 (a workgroup's tail rows belong to the NEXT group and must be left alone; the next group starts in the
    middle of a 32-row block): every replica equals its oracle - per order, counters, list order - through vds_run (graph) and
    through hooked steps."""
    g = load_golden("tiny_kmeans")
    day = synth_days(g, 1, seed=12)
    init = _init(g, R, 500)
    env = mk_env(g, R)
    env.load_orders(*day[0])
    assert env.main_kernel() == "k_tick_dense"
    env.set_run_groups(groups, 0)
    env.reset(init)
    env.run(env.T)
    env.sync()
    _check(env, g, day, np.zeros(R, dtype=np.int32), init)
    ref = env.counters().copy()
    env.reset_again()
    for _ in range(env.T):
        env.step(); env.advance()
    env.sync()
    np.testing.assert_array_equal(env.counters(), ref)
    _check(env, g, day, np.zeros(R, dtype=np.int32), init, replicas=[0, 31, 32, 47, 48, 63, 64, R - 1])
    env.close()


@pytest.mark.parametrize("name", ["tiny_kmeans_dfs2", "tiny_kmeans"])
def test_day_maps_reloaded_on_one_handle_in_groups(name):
    """One handle, grouped runs: an interleaved replica -> day map (replicas stored regrouped by day, padded: more stored replicas
    than the caller's), then a block-wise map (identity storage), then one shared day - the day graph is re-built or updated as the
    shape changes and every replica equals its own oracle each time."""
    g = load_golden(name)
    R = 40
    days = synth_days(g, 3, seed=61)
    init = _init(g, R, 23)
    env = mk_env(g, R)
    maps = [(np.arange(R) % 3).astype(np.int32), np.array([0] * 16 + [2] * 16 + [1] * 8, dtype=np.int32), None, (np.arange(R) % 3).astype(np.int32)]
    for rd in maps:
        if rd is None:
            env.load_orders(*days[1])
            rd, dd = np.zeros(R, dtype=np.int32), [days[1]]
        else:
            env.load_order_days(days, rd)
            dd = days
        env.set_run_groups(3, 1)
        assert env.run_groups() == 3
        env.reset(init)
        env.run(env.T)
        env.sync()
        _check(env, g, dd, rd, init)
    env.close()


def test_bad_arguments_are_refused():
    g = load_golden("tiny_kmeans_dfs2")
    env = mk_env(g, 4)
    with pytest.raises(Exception):
        env.set_run_groups(17, 0)
    with pytest.raises(Exception):
        env.set_run_groups(2, 3)
    env.set_run_groups(0, -1)
    env.close()


def test_large_city_runs_as_two_groups_from_32_replicas_on():
    """The plain tick takes two chains below 256 replicas when the city is large (clusters x replicas >= 192 x 256: configs[4] at
    128 replicas): 1536 clusters x 32 replicas picks 2 groups by itself, 31 replicas of the same city 1; identical results."""
    from vehicles_dispatch_simulator_amd import workloads
    w = workloads.tiny(N=3200, C=1536, vehicles=2500, orders=6000, seed=23)
    outs = []
    for R, groups, want in ((32, 0, 2), (32, 1, 1), (31, 0, 1)):
        env = w.make_env(R)
        assert env.run_groups() == want if groups == 0 else True
        if groups:
            env.set_run_groups(groups, 0)
        assert env.run_groups() == want
        env.reset(w.vehicle_nodes(R))
        env.run(env.T)
        env.sync()
        outs.append((env.counters().copy(), {k: np.array(v) for k, v in env.orders().items()}))
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        np.testing.assert_array_equal(outs[0][1][k], outs[1][1][k], err_msg=k)
        np.testing.assert_array_equal(outs[0][1][k][:31], outs[2][1][k], err_msg=k)
    np.testing.assert_array_equal(outs[0][0][:31], outs[2][0])
