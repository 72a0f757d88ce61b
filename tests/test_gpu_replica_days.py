"""Per-replica order days (vds_load_order_days / vds_load_orders_strided): in the reference one Simulation is one city with
its own self.Orders (simulator.py:325-342), so R replicas may replay R different days with different tick grids.
Every replica is compared with ITS OWN oracle, per tick (counters, observations, container order) and per order, in every
engine mode; a replica whose day is over must stand still; identical days through the per-row kernel variant must equal
the shared-day kernel bit for bit."""
import random

import numpy as np
import pytest

from helpers import engine_settings, load_golden
from oracle.oracle import Oracle
from test_gpu_parity import check_lists
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv, synth, workloads

pytestmark = pytest.mark.gpu

MODES = {"fast": dict(), "generic": dict(force_generic=1), "rows": dict(force_generic=5), "gen2": dict(force_generic=3), "dense16": dict(dense_debug=(16, 0, 0, 0)), "dense8_tiny": dict(dense_debug=(8, 12, 2, 0)), "dense_ring": dict(dense_debug=(16, 0, 0, 2))}


def synth_days(g, n_days, seed):
    """n_days synthetic days on the fixture's city: different order counts, different first / last release (so
    different now0 and tick counts), one of them much shorter."""
    N = int(g["N"])
    days = []
    for d in range(n_days):
        O = [2500, 1800, 700, 3100, 40][d % 5]
        start, pick, dele = synth.make_orders(seed + 17 * d, N, O)
        rel = synth.release_minutes(start)
        if d % 5 == 2:
            keep = rel < rel[0] + 300            # a day that ends after five hours
            rel, pick, dele = rel[keep], pick[keep], dele[keep]
        valid = (g["node2cluster"][pick] >= 0) & (g["node2cluster"][dele] >= 0)
        rel, pick, dele = rel[valid], pick[valid], dele[valid]
        days.append(((rel - rel[0] + 3 * d).astype(np.int32), pick.astype(np.int32), dele.astype(np.int32)))
    return days


def mk_env(g, R, **kw):
    return BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=int(g["V"]),
                              depth_limit=int(g["depth_limit"]), neighbor_can_server=bool(g["neighbor_can_server"]),
                              **{**engine_settings(g), **kw})


def mk_oracle(g, day):
    return Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], int(g["depth_limit"]), bool(g["neighbor_can_server"]),
                  day[0], day[1], day[2], int(g["V"]), **engine_settings(g))


@pytest.mark.parametrize("name,mode", [("tiny_kmeans", "fast"), ("tiny_kmeans", "generic"), ("tiny_grid", "fast"),
                                       # (mixed days inside a group of 16 replicas: the dense tick's per-row order streams, day mode 2)
                                       ("tiny_kmeans", "dense16"), ("tiny_kmeans", "dense8_tiny"), ("tiny_kmeans", "dense_ring"), ("tiny_grid", "dense16"),
                                       ("tiny_grid", "dense8_tiny"),
                                       ("tiny_kmeans", "rows"), ("tiny_kmeans_dfs2", "fast"), ("tiny_kmeans_dfs2", "gen2"), ("tiny_kmeans_dfs2", "generic"),
                                       ("tiny_window6", "fast"), ("tiny_empty_clusters_dfs2", "fast")])
def test_every_replica_replays_its_own_day(name, mode):
    g = load_golden(name)
    V, N, R = int(g["V"]), int(g["N"]), 7
    days = synth_days(g, 4, seed=900 + len(name))
    replica_day = np.array([0, 1, 2, 3, 2, 0, 1], dtype=np.int32)
    valid = g["node2cluster"] >= 0
    init = np.stack([synth.init_vehicle_nodes(random.Random(50 + r), N, V, valid) for r in range(R)]).astype(np.int32)
    init[5] = init[0]                             # same day, same start: must stay identical
    env = mk_env(g, R, **MODES[mode])
    env.load_order_days(days, replica_day)
    if mode == "fast" and bool(g["neighbor_can_server"]) and int(g["depth_limit"]) > 0:
        assert env.main_kernel() == "k_dfs_hybrid"          # per-row order streams in stamp mode: the hybrid tick serves interleaved days too
    env.reset(init)
    oracles = []
    for r in range(R):
        o = mk_oracle(g, days[replica_day[r]])
        o.reset(init[r])
        oracles.append(o)
    Ts = [o.num_ticks for o in oracles]
    assert env.T == max(Ts) and len(set(Ts)) > 1
    for r in range(R):
        assert env.replica_ticks(r) == (Ts[r], days[replica_day[r]][0].size)
    frozen = {}
    for t in range(env.T):
        env.step()
        ob, cn = env.obs(), env.counters()
        for r, o in enumerate(oracles):
            if t < Ts[r]:
                o.begin_tick()
                oo, oc = o.obs(), o.counters()
                for a, b in (("idle_pre", "idle_pre"), ("idle_now", "idle_post"), ("supply", "supply"), ("cl_orders", "cl_orders"), ("inflight", "inflight")):
                    np.testing.assert_array_equal(ob[a][r], oo[b], err_msg="tick %d replica %d obs %s" % (t, r, a))
                for i, k in enumerate(("order_num", "reject_num", "matched", "wait_sum")):
                    assert cn[r, i] == oc[k], (t, r, k, cn[r, i], oc[k])
                assert cn[r, 7] == oc["evals"] and cn[r, 6] == oc["sum_order_value"], (t, r)
                if t % 7 == 0 or t == Ts[r] - 1:
                    check_lists(env, r, o, t)
                o.end_tick()
                if t == Ts[r] - 1:
                    frozen[r] = (env.lists(r), cn[r].copy())
            else:
                # SimCity of this city has returned (:1048): nothing moves any more
                L0, c0 = frozen[r]
                assert (cn[r] == c0).all(), (t, r)
                if t % 5 == 0:
                    L1 = env.lists(r)
                    for k in L0:
                        np.testing.assert_array_equal(L1[k], L0[k], err_msg="replica %d moved after its day ended (%s)" % (r, k))
        env.advance()
    env.sync()
    od = env.orders()
    for r, o in enumerate(oracles):
        exp = o.orders()
        n = exp["status"].size
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(od[k][r][:n], exp[k], err_msg="replica %d %s" % (r, k))
        assert (od["status"][r][n:] == 0).all() and (od["vehicle"][r][n:] == -1).all()
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(od[k][5], od[k][0])
    tot = env.total_counters()
    assert tot[0] == sum(o.counters()["order_num"] for o in oracles) and tot[6] == sum(o.counters()["sum_order_value"] for o in oracles)
    # the same days again (episode restart), through vds_run
    env.reset_again(); env.run(env.T)
    assert (env.total_counters() == tot).all()
    env.close()


@pytest.mark.parametrize("mode", ["fast", "dense16", "dense8_tiny", "dense_ring", "rows"])
@pytest.mark.parametrize("V", [33, 64, 127, 128, 129, 255, 256, 257])      # (order days per replica: 16 lanes per replica, tables of up to 256 entries)
def test_full_tables_rows_with_and_without_an_order_at_each_step(V, mode):
    """Constructed case for the class of bug the round-3 fuzz found by luck (a row WITHOUT an order at a match step, next to rows
    with one, while its register table is FULL, lost the vehicle in the table's last slot): sixteen replicas, every one on its own
    day, all V vehicles parked in one cluster so that its idle list fills the 32- / 64- / 128-entry tables exactly (and one short /
    one beyond), orders that come and go per replica and slot - at every step of the match loop some rows of a wavefront have an
    order and others have none.  Every replica against its own oracle, per tick: counters, observations, list order."""
    g = dict(load_golden("tiny_kmeans"))
    N, R = int(g["N"]), 16
    n2c = g["node2cluster"]
    c0 = int(np.bincount(n2c[n2c >= 0]).argmax())
    nodes = np.flatnonzero(n2c == c0)
    g["V"] = np.int64(V)
    init = np.stack([nodes[(np.arange(V) * (r + 1) + r) % nodes.size] for r in range(R)]).astype(np.int32)
    days = []
    for r in range(R):
        rel, pk, dl = [], [], []
        for t in range(14):
            n = 0 if (t + r) % 4 == 0 else 1 + (t * 7 + r * 3) % 6
            if r == 5 and t >= 6: n = 0                      # a day that ends early: its row stands still
            for i in range(n):
                rel.append(10 * t + (i % 10)); pk.append(int(nodes[(3 * t + 5 * i + r) % nodes.size])); dl.append(int(nodes[(7 * t + i + 2 * r) % nodes.size]))
        rel.append(rel[-1] + 1); pk.append(int(nodes[0])); dl.append(int(nodes[0]))      # (the last order is never processed, :914)
        days.append((np.array(rel, np.int32), np.array(pk, np.int32), np.array(dl, np.int32)))
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=V, depth_limit=0, neighbor_can_server=False,
                             **{**engine_settings(g), **MODES[mode]})
    env.load_order_days(days, np.arange(R, dtype=np.int32))
    assert env.main_kernel() == ("k_tick_rows" if mode == "rows" else "k_tick_dense")
    env.reset(init)
    oracles = []
    for r in range(R):
        o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], 0, False, days[r][0], days[r][1], days[r][2], V, **engine_settings(g))
        o.reset(init[r]); oracles.append(o)
    for t in range(env.T):
        env.step()
        ob, cn = env.obs(), env.counters()
        for r, o in enumerate(oracles):
            if t >= o.num_ticks: continue
            o.begin_tick()
            oo, oc = o.obs(), o.counters()
            for a, b in (("idle_pre", "idle_pre"), ("idle_now", "idle_post"), ("supply", "supply"), ("cl_orders", "cl_orders"), ("inflight", "inflight")):
                np.testing.assert_array_equal(ob[a][r], oo[b], err_msg="tick %d replica %d obs %s" % (t, r, a))
            assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["evals"]), (t, r)
            check_lists(env, r, o, t)
            o.end_tick()
        env.advance()
    env.sync()
    od = env.orders()
    for r, o in enumerate(oracles):
        exp = o.orders()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(od[k][r][:exp[k].size], exp[k], err_msg="replica %d %s" % (r, k))
    env.close()


def test_strided_form_and_identical_days_equal_the_shared_day_kernel():
    g = load_golden("tiny_kmeans")
    V, N, R = int(g["V"]), int(g["N"]), 6
    valid = g["node2cluster"] >= 0
    init = np.stack([synth.init_vehicle_nodes(random.Random(70 + r), N, V, valid) for r in range(R)]).astype(np.int32)
    day = (g["o_release_min"], g["o_pickup"], g["o_delivery"])
    O = day[0].size
    ref = mk_env(g, R); ref.load_orders(*day); ref.reset(init); ref.run(ref.T)
    assert ref.main_kernel() == "k_tick_dense"
    exp, expc = ref.orders(), ref.counters()
    # (a) R copies of the day, stride O: every row goes through the per-row variant
    a = mk_env(g, R)
    a.load_orders_strided(np.tile(day[0], R), np.tile(day[1], R), np.tile(day[2], R), O, O)
    a.reset(init); a.run(a.T)
    # (b) padded stride, (c) stride 0 == shared day
    pad = 13
    rows = [np.concatenate([np.tile(np.concatenate([x, np.zeros(pad, np.int32)]), R)]) for x in day]
    b = mk_env(g, R); b.load_orders_strided(rows[0], rows[1], rows[2], O, O + pad); b.reset(init); b.run(b.T)
    c = mk_env(g, R); c.load_orders_strided(day[0], day[1], day[2], O, 0); c.reset(init); c.run(c.T)
    for env in (a, b, c):
        got = env.orders()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k], exp[k])
        assert (env.counters() == expc).all()
        env.close()
    with pytest.raises(Exception, match="replica_stride"):
        d = mk_env(g, R); d.load_orders_strided(day[0], day[1], day[2], O, O - 1)
    ref.close()


def test_full_size_distinct_days_config2():
    """configs[1] at bench size with 8 distinct synthetic days over 1024 replicas: anchors against the oracle."""
    R, D = 1024, 8
    w = workloads.didi_day("cfg2")
    days = workloads.distinct_days(w, D)
    init = w.vehicle_nodes(R)
    env = w.make_env(R, load=False)
    env.load_order_days(days, np.arange(R) % D)         # interleaved: every row of a wavefront on another day
    env.reset(init)
    env.run(env.T)
    env.sync()
    cn = env.counters()
    for r in (0, 1, 7, 8, 515, 1023):
        d = days[r % D]
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, d[0], d[1], d[2], w.vehicles)
        o.reset(init[r]); o.run_day()
        exp, oc, got = o.orders(), o.counters(), env.orders(r, 1)
        n = exp["status"].size
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][0][:n], exp[k], err_msg="replica %d %s" % (r, k))
        assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
    # the default map (contiguous blocks of R / D = 128 replicas per day: one day per workgroup): same days, other replicas
    env.load_order_days(days)
    env.reset(init)
    env.run(env.T)
    env.sync()
    cn = env.counters()
    for r in (0, 127, 128, 600, 1023):
        d = days[r * D // R]
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, d[0], d[1], d[2], w.vehicles)
        o.reset(init[r]); o.run_day()
        exp, oc, got = o.orders(), o.counters(), env.orders(r, 1)
        n = exp["status"].size
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][0][:n], exp[k], err_msg="block map: replica %d %s" % (r, k))
        assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["evals"])
    env.close()


@pytest.mark.parametrize("layout", ["interleaved", "blocked"])
def test_full_size_distinct_days_config4_neighbour_search(layout):
    """configs[3] (neighbour DFS depth 2, the hybrid tick) at bench size with 8 distinct days over 1024 replicas: 16 replicas
    - every day twice, batch positions far apart - against the oracle of THEIR day (per-order results, counters), and for ALL
    1024 replicas the size-independent invariants: rejects + matched = processed orders of the replica's day, every vehicle
    either idle or in flight, identical results for replicas with the same day and the same start nodes."""
    R, D = 1024, 8
    w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
    days = workloads.distinct_days(w, D)
    init = w.vehicle_nodes(R)
    rmap = (np.arange(R) % D) if layout == "interleaved" else (np.arange(R) * D // R)
    twin = int(np.flatnonzero((rmap == rmap[777]) & (np.arange(R) != 777))[5])
    init[777] = init[twin]
    env = w.make_env(R, load=False)
    env.load_order_days(days, rmap.astype(np.int32))
    assert env.main_kernel() == "k_dfs_hybrid"
    env.reset(init)
    env.run(env.T)
    env.sync()
    cn = env.counters()
    ob = env.obs()
    # invariants, every replica
    proc = np.array([int(env.replica_ticks(r)[1]) for r in range(R)])
    assert np.array_equal(cn[:, 0], cn[:, 1] + cn[:, 2])
    assert (cn[:, 0] <= proc).all() and (cn[:, 0] >= proc - 1).all()          # the last order is never processed (:914-915)
    assert np.array_equal(ob["idle_now"].sum(axis=1) + ob["inflight"].sum(axis=1), np.full(R, w.vehicles))
    np.testing.assert_array_equal(cn[777], cn[twin])
    a, b2 = env.orders(777, 1), env.orders(twin, 1)
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(a[k], b2[k])
    # oracle anchors: each day twice
    picks = []
    for d in range(D):
        rr = np.flatnonzero(rmap == d)
        picks += [int(rr[0]), int(rr[-1])]
    for r in picks:
        d = days[rmap[r]]
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, d[0], d[1], d[2], w.vehicles)
        o.reset(init[r]); o.run_day()
        exp, oc, got = o.orders(), o.counters(), env.orders(r, 1)
        n = exp["status"].size
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][0][:n], exp[k], err_msg="%s: replica %d %s" % (layout, r, k))
        assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
    env.close()


@pytest.mark.parametrize("name,layout", [("tiny_kmeans", "blocks"), ("tiny_grid", "blocks"), ("tiny_kmeans", "interleaved"),
                                         ("tiny_kmeans_dfs2", "interleaved"), ("tiny_kmeans_dfs2", "blocks")])
def test_days_in_blocks_of_sixteen_replicas_take_the_workgroup_uniform_path(name, layout):
    """A block-wise replica -> day map (every aligned group of 16 replicas on one day) runs the shared-day kernel code with
    the day looked up once per workgroup (k_tick_rows day mode 1): every replica still equals its own oracle, incl. a
    last, partially filled group and a day that ends early.  "interleaved" (replica r on day r % 3, 14 / 13 / 13 replicas):
    the library forms the workgroups from the replicas of one day itself (row slot -> replica permutation, padded groups)."""
    g = load_golden(name)
    V, N, R = int(g["V"]), int(g["N"]), 40
    days = synth_days(g, 3, seed=333)
    replica_day = np.array([0] * 16 + [2] * 16 + [1] * 8, dtype=np.int32) if layout == "blocks" else (np.arange(R) % 3).astype(np.int32)
    valid = g["node2cluster"] >= 0
    init = np.stack([synth.init_vehicle_nodes(random.Random(90 + r), N, V, valid) for r in range(R)]).astype(np.int32)
    env = mk_env(g, R)
    env.load_order_days(days, replica_day)
    env.reset(init)
    env.run(env.T)
    env.sync()
    got, cn = env.orders(), env.counters()
    for r in range(R):
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
    env.close()


def test_reloading_days_on_one_handle_neighbour_search():
    """vds_load_orders again and again on one handle (Simulation.Reload): the per-day tables of the hybrid neighbour-search tick
    (visit rows, cost bounds, steal log) are rebuilt each time; every day equals its oracle, incl. the evaluations."""
    g = load_golden("tiny_kmeans_dfs2")
    V, N, R = int(g["V"]), int(g["N"]), 5
    days = synth_days(g, 3, seed=77)
    valid = g["node2cluster"] >= 0
    init = np.stack([synth.init_vehicle_nodes(random.Random(5 + r), N, V, valid) for r in range(R)]).astype(np.int32)
    env = mk_env(g, R)
    for d in (0, 1, 2, 0):
        env.load_orders(*days[d])
        assert env.main_kernel() == "k_dfs_dense"
        env.reset(init)
        env.run(env.T)
        env.sync()
        got, cn = env.orders(), env.counters()
        for r in range(R):
            o = mk_oracle(g, days[d])
            o.reset(init[r]); o.run_day()
            exp, oc = o.orders(), o.counters()
            for k in ("status", "vehicle", "wait"):
                np.testing.assert_array_equal(got[k][r][:exp[k].size], exp[k], err_msg="day %d replica %d %s" % (d, r, k))
            assert cn[r, 7] == oc["evals"] and cn[r, 1] == oc["reject_num"]
    env.close()


def test_dispatch_on_a_replica_whose_day_is_over_is_refused():
    """A replica past its last slot stands still (``:1048``): moving one of its vehicles would strand it (nobody drains the
    arrival tables of a finished city).  The host-list call refuses the whole call, the device-tensor call skips that
    replica's actions and reports it at the next sync; live replicas are served in both."""
    import torch
    g = load_golden("tiny_kmeans")
    V, N, R = int(g["V"]), int(g["N"]), 4
    days = synth_days(g, 3, seed=77)            # day 2 ends after five hours
    replica_day = np.array([0, 2, 1, 2], dtype=np.int32)
    valid = g["node2cluster"] >= 0
    init = np.stack([synth.init_vehicle_nodes(random.Random(9 + r), N, V, valid) for r in range(R)]).astype(np.int32)
    env = mk_env(g, R, stream=torch.cuda.current_stream().cuda_stream)
    env.load_order_days(days, replica_day)
    env.reset(init)
    T_short = env.replica_ticks(1)[0]
    assert T_short < env.T - 3
    for _ in range(T_short + 2):
        env.step(); env.advance()
    env.step()
    ob = env.obs()
    tgt = int(np.flatnonzero(valid)[0])
    cl_live = int(np.argmax(ob["idle_now"][0])); cl_done = int(np.argmax(ob["idle_now"][1]))
    before = env.counters().copy()
    with pytest.raises(Exception, match="day of replica 1 is over"):
        env.apply_dispatch([0, 1], [cl_live, cl_done], [0, 0], [tgt, tgt])
    np.testing.assert_array_equal(env.counters(), before)            # nothing of the refused call was applied
    env.apply_dispatch([0], [cl_live], [0], [tgt])
    acts = np.full((R, 2, 3), -1, dtype=np.int32)
    acts[2, 0] = (int(np.argmax(ob["idle_now"][2])), 0, tgt)
    acts[3, 1] = (int(np.argmax(ob["idle_now"][3])), 0, tgt)           # replica 3 replays the short day too
    env.apply_dispatch_torch(torch.from_numpy(acts).cuda())
    with pytest.raises(Exception, match="dispatch"):
        env.sync()
    cn = env.counters()
    assert cn[0, 4] == before[0, 4] + 1 and cn[2, 4] == before[2, 4] + 1 and cn[1, 4] == before[1, 4] and cn[3, 4] == before[3, 4]
    assert env.obs()["inflight"][3].sum() == ob["inflight"][3].sum()
    env.close()


def test_loader_wrappers_check_array_lengths():
    g = load_golden("tiny_kmeans")
    env = mk_env(g, 3)
    rel, pk, dl = g["o_release_min"], g["o_pickup"], g["o_delivery"]
    O = rel.size
    with pytest.raises(Exception, match="equal length"):
        env.load_order_days([(rel, pk, dl), (rel, pk[:-1], dl)])
    with pytest.raises(Exception, match="need .* elements"):
        env.load_orders_strided(np.tile(rel, 2), np.tile(pk, 2), np.tile(dl, 2), O, O)      # 3 replicas need 3 * O
    env.close()


@pytest.mark.parametrize("name,R", [("tiny_kmeans", 40), ("tiny_kmeans_dfs2", 40), ("tiny_grid", 7)])
def test_replica_day_map_changes_every_episode_over_resident_days(name, R):
    """vds_set_replica_days: the order days stay resident, only the replica -> day map changes between episodes (the reference: one
    Reload per city, simulator.py:130-212).  Maps of every storage form follow each other on ONE handle - days in aligned blocks of 16
    (one day per workgroup), interleaved (stored regrouped by day, with padding replicas), random with few replicas per day (one order
    stream per row), everything on one day - and every replica is compared with the oracle of ITS day after every episode."""
    g = load_golden(name)
    V, N = int(g["V"]), int(g["N"])
    days = synth_days(g, 5 if R == 40 else 4, seed=4100 + R)
    rng = np.random.default_rng(31 + R)
    valid = g["node2cluster"] >= 0
    init = np.stack([synth.init_vehicle_nodes(random.Random(7000 + r), N, V, valid) for r in range(R)])
    env = mk_env(g, R)
    maps = [np.minimum(np.arange(R) // 16, 3), np.arange(R) % 4, rng.integers(0, 4, size=R), np.arange(R) % 2, np.zeros(R, dtype=np.int64), (np.arange(R) * 7 + 1) % 4,
            np.minimum(np.arange(R) // 16, 3)[::-1].copy()]
    if R == 40:
        # day groups of EIGHT replicas (the dense tick's 8-row workgroups): as given, and interleaved over five days - regrouped by
        # day in groups of eight (groups of sixteen would double the replicas) - then back to one stream per row and to blocks of 16
        maps += [np.arange(R) // 8, np.arange(R) % 5, rng.integers(0, 5, size=R), (np.arange(R) // 8 + 2) % 5, np.minimum(np.arange(R) // 16, 3),
                 (np.arange(R) // 4) % 5, np.arange(R) % 5]       # (... and of four: one wavefront per workgroup)
    env.load_order_days(days, maps[0].astype(np.int32))
    expected = {}
    again = 0
    for ep, rd in enumerate(maps):
        if ep:
            env.set_replica_days(rd.astype(np.int32))
            with pytest.raises(Exception, match="reset"):
                env.step()                                  # the episode state is void until a reset
        # every other episode keeps its start nodes: vds_reset_again is a valid follow-up of vds_set_replica_days while the state tables
        # were not re-allocated (the map changed the number of stored replicas: padding) - then only vds_reset / vds_reset_random are
        if ep % 2 == 1:
            try:
                env.reset_again()
                again += 1
            except Exception as e:
                assert "re-allocated" in str(e), e
                env.reset(init)
        else:
            env.reset(init)
        assert env.T == max(mk_oracle(g, days[int(d)]).num_ticks for d in set(rd.tolist()))
        env.run(env.T)
        got, cn = env.orders(), env.counters()
        for r in range(R):
            d = int(rd[r])
            key = (r, d)
            if key not in expected:
                o = mk_oracle(g, days[d]); o.reset(init[r]); o.run_day()
                expected[key] = (o.orders(), o.counters())
            exp, oc = expected[key]
            n = exp["status"].size
            for k in ("status", "vehicle", "wait"):
                np.testing.assert_array_equal(got[k][r][:n], exp[k], err_msg="episode %d replica %d day %d %s" % (ep, r, d, k))
            assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["evals"]), (ep, r)
            assert env.replica_ticks(r)[0] == mk_oracle(g, days[d]).num_ticks if hasattr(env, "replica_ticks") else True
    assert again >= 2
    with pytest.raises(Exception, match="mapped to day"):
        env.set_replica_days(np.full(R, 9, dtype=np.int32))
    env.close()
