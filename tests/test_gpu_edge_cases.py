"""Edge cases of the tick path through the C ABI, each against the CPU oracle (bit-exact):
small pickup window (window rejects, quirk Q3 made live), unsorted release times (cursor semantics),
no vehicles, negative / huge costs (generic kernels), API misuse."""
import random

import numpy as np
import pytest

from helpers import load_golden
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv, synth
from vehicles_dispatch_simulator_amd import _lib as _lib_mod

pytestmark = pytest.mark.gpu


def run_both(g, R, V, threshold=600_000_000_000, nbr=None, depth=None, release=None, cost=None, init=None, **kw):
    cost = g["cost"] if cost is None else cost
    release = g["o_release_min"] if release is None else release
    ncs = bool(g["neighbor_can_server"]) if nbr is None else nbr
    depth = int(g["depth_limit"]) if depth is None else depth
    N = int(g["N"])
    if init is None:
        init = np.stack([synth.init_vehicle_nodes(random.Random(5 + r), N, V, g["node2cluster"] >= 0) for r in range(R)]) if V else np.zeros((R, 0), np.int32)
    env = BatchedDispatchEnv(cost, g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=V, depth_limit=depth,
                             neighbor_can_server=ncs, reject_threshold=threshold, **kw)
    env.load_orders(release, g["o_pickup"], g["o_delivery"])
    env.reset(init)
    env.run(env.T)
    got, cn = env.orders(), env.counters()
    obs = env.obs()
    for r in range(R):
        o = Oracle(cost, g["node2cluster"], g["nbr_off"], g["nbr_idx"], depth, ncs, release, g["o_pickup"], g["o_delivery"], V,
                   reject_threshold=threshold)
        o.reset(init[r])
        assert o.run_day() == env.T
        exp, oc = o.orders(), o.counters()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][r], exp[k], err_msg="replica %d %s" % (r, k))
        for i, k in enumerate(("order_num", "reject_num", "matched", "wait_sum")):
            assert cn[r, i] == oc[k], (r, k)
        assert cn[r, 6] == oc["sum_order_value"] and cn[r, 7] == oc["evals"]
        np.testing.assert_array_equal(obs["idle_now"][r], o.obs()["idle_now"])
    env.close()
    return cn


@pytest.mark.parametrize("name", ["tiny_kmeans", "tiny_kmeans_dfs2", "tiny_grid_nbr_scarce"])
@pytest.mark.parametrize("threshold", [0, 3, 12])
def test_small_pickup_window_rejects(name, threshold):
    """cost > threshold rejects WITHOUT consuming a vehicle (:943): exercises the generic kernels and, with
    neighbour search, the 'not dry after all' branch of the lower-bound rounds."""
    g = load_golden(name)
    cn = run_both(g, R=5, V=int(g["V"]), threshold=threshold)
    assert cn[:, 1].min() > 0


@pytest.mark.parametrize("name", ["tiny_kmeans", "tiny_kmeans_dfs1"])
def test_unsorted_release_times_follow_cursor_semantics(name):
    g = load_golden(name)
    rel = g["o_release_min"].copy()
    rng = np.random.default_rng(3)
    idx = rng.choice(rel.size - 1, size=200, replace=False)
    rel[idx] = np.maximum(0, rel[idx] + rng.integers(-300, 300, size=200)).astype(np.int32)   # some early, some late
    run_both(g, R=3, V=int(g["V"]), release=rel)


def test_no_vehicles_everything_rejected():
    g = load_golden("tiny_kmeans")
    cn = run_both(g, R=2, V=0)
    assert (cn[:, 1] == cn[:, 0]).all() and cn[0, 0] == g["o_pickup"].size - 1


@pytest.mark.parametrize("name", ["tiny_kmeans", "tiny_kmeans_dfs2"])
def test_negative_and_huge_costs_use_generic_kernels(name):
    g = load_golden(name)
    cost = g["cost"].astype(np.int64).copy()
    rng = np.random.default_rng(1)
    mask = rng.random(cost.shape) < 0.02
    cost[mask] = rng.integers(-50, 0, size=int(mask.sum()))
    mask2 = rng.random(cost.shape) < 0.01
    cost[mask2] = 99999                      # the reference's empty-cluster sentinel magnitude
    run_both(g, R=3, V=int(g["V"]), cost=cost.astype(np.int32), far_cap=256)


def test_single_replica_single_cluster_city():
    g = load_golden("tiny_kmeans")
    g = dict(g)
    g["node2cluster"] = np.zeros_like(g["node2cluster"])
    g["nbr_off"] = np.zeros(2, np.int32)
    g["nbr_idx"] = np.zeros(0, np.int32)
    run_both(g, R=1, V=90, idle_cap=128, ring_cap=128)


def test_api_misuse_is_reported_not_fatal():
    g = load_golden("tiny_kmeans")
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=2, vehicles=10)
    with pytest.raises(Exception, match="vds_reset"):
        env.step()
    env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
    with pytest.raises(Exception, match="in no cluster|start nodes"):
        env.reset(np.full((2, 10), 10**6, np.int32))
    env.reset(np.zeros((2, 10), np.int32))
    with pytest.raises(Exception, match="must follow vds_step"):
        env.apply_dispatch([0], [0], [0], [0])
    env.step()
    with pytest.raises(Exception, match="already stepped"):
        env.step()
    c0 = int(g["node2cluster"][0])
    env.apply_dispatch([0], [c0], [500], [1])          # no such idle position
    with pytest.raises(Exception, match="dispatch of an idle position"):
        env.sync()
    env.reset(np.zeros((2, 10), np.int32))              # reset clears the sticky error
    env.run(env.T)
    with pytest.raises(Exception, match="past the end of the day"):
        env.step()
    with pytest.raises(Exception, match="replica range"):
        env.orders(1, 5)
    with pytest.raises(Exception, match="replica out of range"):
        env.vehicles(7)
    import ctypes as C
    assert env._lib.vds_apply_dispatch_device(env._h, 65, C.c_void_p(1)) != 0          # K > 64
    assert env._lib.vds_apply_dispatch_device(env._h, 4, None) != 0                    # no tensor
    assert env._lib.vds_counters_device(env._h, None) != 0
    assert env._lib.vds_main_kernel(env._h) == b"k_tick_dense"
    env.close()


def test_handles_on_their_own_streams_advance_independently():
    """INTEGRATION.md "different days in parallel": several handles on one GPU, each on its own HIP stream
    (torch side streams adopted through vds_set_stream), ticks issued interleaved without host syncs."""
    import torch
    names = ["tiny_kmeans", "tiny_kmeans_dfs2", "tiny_grid", "tiny_tick5"]
    envs, inits, gs, streams = [], [], [], []
    for i, name in enumerate(names):
        g = load_golden(name)
        R, V, N = 5 + i, int(g["V"]), int(g["N"])
        st = torch.cuda.Stream()
        init = np.stack([g["veh_node"]] + [synth.init_vehicle_nodes(random.Random(40 + 10 * i + r), N, V, g["node2cluster"] >= 0) for r in range(1, R)])
        env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=V,
                                 depth_limit=int(g["depth_limit"]), neighbor_can_server=bool(g["neighbor_can_server"]),
                                 tick_minutes=int(g["tick_minutes"]) if "tick_minutes" in g else 10, stream=st.cuda_stream)
        env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
        env.reset(init)
        envs.append(env); inits.append(init); gs.append(g); streams.append(st)
    for t in range(max(e.T for e in envs)):
        for e in envs:
            if t < e.T:
                e.step()
                e.advance()
    for e, g, init in zip(envs, gs, inits):
        got = e.orders()
        np.testing.assert_array_equal(got["vehicle"][0], g["o_vehicle"])      # replica 0 = the reference's own day
        np.testing.assert_array_equal(got["wait"][0], g["o_wait"])
        for r in range(1, e.R):
            o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], int(g["depth_limit"]), bool(g["neighbor_can_server"]),
                       g["o_release_min"], g["o_pickup"], g["o_delivery"], int(g["V"]),
                       tick_minutes=int(g["tick_minutes"]) if "tick_minutes" in g else 10)
            o.reset(init[r]); o.run_day()
            exp = o.orders()
            for k in ("status", "vehicle", "wait"):
                np.testing.assert_array_equal(got[k][r], exp[k], err_msg="%d %s" % (r, k))
        e.close()


def test_legacy_default_stream_can_be_bound_explicitly():
    """vds_set_stream(VDS_STREAM_LEGACY_DEFAULT): engine launches on hipStream_t 0 itself - PyTorch's default stream on ROCm - so a
    policy's tensor operations and the engine's kernels are ordered by one stream.  vds_run (stream capture refuses that stream:
    the day is issued launch by launch), hooked steps with a torch-side read in between, and the results equal the oracle's."""
    import torch
    g = load_golden("tiny_kmeans")
    R, V, N = 5, int(g["V"]), int(g["N"])
    init = np.stack([g["veh_node"]] + [synth.init_vehicle_nodes(random.Random(90 + r), N, V, g["node2cluster"] >= 0) for r in range(1, R)])
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=V, stream="legacy")
    env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
    env.reset(init)
    seen = []
    for t in range(env.T // 2):
        env.step()
        seen.append(env.obs_torch(inflight=False)[1].sum())      # a torch op on the default stream, right behind the tick: no sync in between
        env.advance()
    env.run(env.T - env.T // 2)
    env.sync()
    got = env.orders()
    np.testing.assert_array_equal(got["vehicle"][0], g["o_vehicle"])
    np.testing.assert_array_equal(got["wait"][0], g["o_wait"])
    assert int(torch.stack(seen).sum().item()) > 0
    env.set_stream(None)                                         # back to the library's own stream: another day, same results
    env.reset_again(); env.run(env.T); env.sync()
    again = env.orders()
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(again[k], got[k])
    env.close()


def test_another_day_on_the_same_handle():
    """vds_load_orders again (Reload, simulator.py:130-212): the day changes, the static tables and - while they
    still fit - the state tables stay; results equal those of a fresh handle / the oracle, in both directions."""
    g = load_golden("tiny_kmeans")
    R, V, N = 3, int(g["V"]), int(g["N"])
    init = np.stack([synth.init_vehicle_nodes(random.Random(90 + r), N, V) for r in range(R)])
    start, pick2, dele2 = synth.make_orders(4242, N, 3500)            # a longer, different day for the same city
    rel2 = synth.release_minutes(start)
    days = [(g["o_release_min"], g["o_pickup"], g["o_delivery"]), (rel2, pick2, dele2), (g["o_release_min"], g["o_pickup"], g["o_delivery"])]
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=V)
    first = None
    for d, (rel, pk, dl) in enumerate(days):
        env.load_orders(rel, pk, dl)
        with pytest.raises(Exception, match="vds_reset"):
            env.step()
        env.reset(init)
        env.run(env.T)
        got, cn = env.orders(), env.counters()
        for r in range(R):
            o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], 0, False, rel, pk, dl, V)
            o.reset(init[r]); assert o.run_day() == env.T
            exp, oc = o.orders(), o.counters()
            for k in ("status", "vehicle", "wait"):
                np.testing.assert_array_equal(got[k][r], exp[k], err_msg="day %d replica %d %s" % (d, r, k))
            assert cn[r, 0] == oc["order_num"] and cn[r, 1] == oc["reject_num"] and cn[r, 3] == oc["wait_sum"] and cn[r, 6] == oc["sum_order_value"]
        if d == 0:
            first = got["vehicle"].copy()
    np.testing.assert_array_equal(got["vehicle"], first)      # day 1 again after day 2: identical
    env.close()


def test_round2_entry_points_refuse_bad_arguments():
    """vds_load_order_days / vds_load_orders_strided / vds_apply_dispatch_ex / vds_set_idle_cap / vds_cluster_cost_sums:
    status code + message, nothing aborts, the handle stays usable."""
    import ctypes as C
    g = load_golden("tiny_kmeans")
    R, V, N = 4, int(g["V"]), int(g["N"])
    day = (g["o_release_min"], g["o_pickup"], g["o_delivery"])
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=V)
    with pytest.raises(Exception, match="load static tables and orders first"):
        env.set_idle_cap(256)
    with pytest.raises(Exception, match="mapped to day"):
        env.load_order_days([day, day], replica_day=[0, 1, 2, 0])
    with pytest.raises(Exception, match="replica_day needs"):
        env.load_order_days([day, day], replica_day=[0, 1])
    with pytest.raises(Exception, match="no orders"):
        env.load_order_days([day, (day[0][:0], day[1][:0], day[2][:0])])
    bad = (day[0], day[1].copy(), day[2]); bad[1][5] = N + 3
    with pytest.raises(Exception, match="node outside"):
        env.load_order_days([day, bad])
    with pytest.raises(Exception, match="replica_stride"):
        env.load_orders_strided(np.tile(day[0], R), np.tile(day[1], R), np.tile(day[2], R), day[0].size, 5)
    with pytest.raises(Exception, match="call vds_reset first|vds_reset"):
        env.load_order_days([day, day]); env.apply_dispatch([0], [0], [0], [0], arrive_min=[5], counted=[0])
    # the handle is still usable
    init = np.stack([synth.init_vehicle_nodes(random.Random(3 + r), N, V) for r in range(R)])
    env.reset(init)
    env.step()
    with pytest.raises(Exception, match="target node"):
        env.apply_dispatch([0], [0], [0], [N + 1], arrive_min=[5], counted=[1])
    with pytest.raises(Exception, match="bad capacity"):
        env.set_idle_cap(0)
    env.advance()
    env.run(env.T - 1); env.sync()
    assert env.counters()[0, 0] == day[0].size - 1
    env.close()
    lib = _lib_mod.load()
    sums = np.zeros((3, 3), np.int64)
    cost = np.zeros((4, 4), np.int32); n2c = np.zeros(4, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.vds_cluster_cost_sums(0, None, 4, p(n2c), 3, p(sums), None) == -1
    assert lib.vds_cluster_cost_sums(99, p(cost), 4, p(n2c), 3, p(sums), None) == -1
    assert b"bad device" in lib.vds_cluster_cost_sums_error()


def make_env(g, R, **kw):
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=int(g["V"]), **kw)
    env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
    return env


@pytest.mark.parametrize("fg", [0, 5])
def test_slow_path_buckets_are_counted(fg):
    """vds_read_work[6]: buckets the fast kernel hands to a slower path.  None on an ordinary day; every tick of the crowded
    cluster when 600 vehicles sit in one cluster (beyond the register tables of both fast kernels)."""
    g = load_golden("tiny_kmeans")
    env = make_env(g, 2, force_generic=fg)
    env.reset(np.tile(g["veh_node"], (2, 1)))
    env.run(env.T)
    assert env.work()["slow_path_buckets"] == 0 and env.main_kernel() == ("k_tick_rows" if fg == 5 else "k_tick_dense")
    env.close()
    g = dict(g); g["V"] = np.int64(600)
    nodes5 = np.flatnonzero(g["node2cluster"] == 5)
    init = nodes5[np.arange(600) % nodes5.size].astype(np.int32)[None, :].repeat(2, axis=0)
    env = make_env(g, 2, idle_cap=640, force_generic=fg)
    env.reset(init)
    env.run(20)
    assert env.work()["slow_path_buckets"] >= 2 * 15
    env.close()


def test_dense_tick_switches_to_wide_rows_when_the_slow_path_is_crowded(monkeypatch):
    """k_tick_dense with one shared order day starts with 8 lanes per replica and 128-entry tables; when an episode hands more
    than 0.1 % of its bucket-ticks to the slow path (here: 200 vehicles parked in one cluster of every replica) the handle takes
    16 lanes per replica and 256-entry tables from the next-but-one episode start on (the counter travels through pinned host
    memory, no synchronisation).  Every episode equals the oracle, before and after the switch; after it the crowded cluster is on
    the fast path (no slow-path buckets).  VDS_DENSE_ADAPT=0 keeps the first form."""
    g = dict(load_golden("tiny_kmeans")); g["V"] = np.int64(200)
    R = 20
    nodes5 = np.flatnonzero(g["node2cluster"] == 5)
    init = np.stack([nodes5[(np.arange(200) * (r + 1)) % nodes5.size] for r in range(R)]).astype(np.int32)
    exp = []
    for r in range(R):
        o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], 0, False, g["o_release_min"], g["o_pickup"], g["o_delivery"], 200)
        o.reset(init[r]); o.run_day(); exp.append(o.orders())
    for adapt in ("1", "0"):
        monkeypatch.setenv("VDS_DENSE_ADAPT", adapt)
        env = make_env(g, R, idle_cap=256)
        assert env.main_kernel() == "k_tick_dense"
        env.reset(init)
        slow = []
        for ep in range(5):
            if ep: env.reset_again()
            env.run(env.T); env.sync()
            got = env.orders()
            for r in range(R):
                for k in ("status", "vehicle", "wait"):
                    np.testing.assert_array_equal(got[k][r], exp[r][k], err_msg="episode %d replica %d %s" % (ep, r, k))
            slow.append(env.work()["slow_path_buckets"])
        assert slow[0] > 0 and slow[1] == slow[0]
        if adapt == "1": assert slow[-1] == 0 and slow[-2] == 0, slow
        else: assert slow[-1] == slow[0], slow
        env.close()


@pytest.mark.parametrize("fg", [0, 5])
@pytest.mark.parametrize("shape", [(300, 12, 150, 3), (300, 12, 1, 2), (700, 37, 5000, 5), (4139, 192, 10000, 3)])
def test_reset_builds_the_same_lists_staged_and_unstaged(shape, fg, monkeypatch):
    """InitVehiclesIntoCluster (:249-258): every cluster's idle list = its vehicles in vehicle order.  k_reset_staged (lists built
    in LDS, written as runs) and k_reset_fast (VDS_RESET_UNSTAGED=1: every entry stored where it belongs) against numpy, on the
    dense (force_generic 0) and the wide layout (5); then a day from each, identical results."""
    from vehicles_dispatch_simulator_amd import workloads
    N, Cn, V, R = shape
    w = workloads.tiny(N=N, C=Cn, vehicles=V, orders=1500, seed=11 + V)
    init = w.vehicle_nodes(R)
    n2c = np.asarray(w.city.node2cluster)
    outs = []
    for unstaged in ("0", "1"):
        monkeypatch.setenv("VDS_RESET_UNSTAGED", unstaged)
        env = w.make_env(R, force_generic=fg)
        env.reset(init)
        for r in range(R):
            ls = env.lists(r)
            cl = n2c[init[r]]
            for c in range(Cn):
                exp = np.nonzero(cl == c)[0]
                a, b = ls["idle_off"][c], ls["idle_off"][c + 1]
                np.testing.assert_array_equal(ls["idle_veh"][a:b], exp, err_msg="replica %d cluster %d" % (r, c))
                np.testing.assert_array_equal(ls["idle_node"][a:b], init[r][exp])
        env.run(env.T)
        env.reset_again()
        env.run(env.T)
        outs.append((env.orders(), env.counters()))
        env.close()
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(outs[0][0][k], outs[1][0][k])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("fg", [0, 5])
def test_reset_again_of_a_city_too_large_for_the_staged_reset_copies_the_list_image(fg, monkeypatch):
    """configs[4]'s reset form at a test size: 30 000 vehicles in 37 clusters do not fit k_reset_staged's LDS.  On the dense layout
    (force_generic 0) the first reset (k_reset_fast) leaves a packed image of the lists (k_reset_capture) and vds_reset_again copies
    it back (k_reset_image); new start nodes and another idle capacity make the image stale.  Lists against
    numpy after every reset; the days against the ones of k_reset_fast alone (VDS_RESET_UNSTAGED=1, and the wide layout)."""
    from vehicles_dispatch_simulator_amd import workloads
    N, Cn, V, R = 700, 37, 30000, 2
    w = workloads.tiny(N=N, C=Cn, vehicles=V, orders=1500, seed=5)
    init = w.vehicle_nodes(R)
    init_b = w.vehicle_nodes(R, first_replica=7)
    n2c = np.asarray(w.city.node2cluster)

    def check_lists(env, nodes):
        for r in range(R):
            ls = env.lists(r)
            cl = n2c[nodes[r]]
            for c in range(Cn):
                exp = np.nonzero(cl == c)[0]
                a, b = ls["idle_off"][c], ls["idle_off"][c + 1]
                np.testing.assert_array_equal(ls["idle_veh"][a:b], exp, err_msg="replica %d cluster %d" % (r, c))
                np.testing.assert_array_equal(ls["idle_node"][a:b], nodes[r][exp])

    outs = []
    for unstaged in ("0", "1"):
        monkeypatch.setenv("VDS_RESET_UNSTAGED", unstaged)
        env = w.make_env(R, force_generic=fg)
        res = []
        env.reset(init); check_lists(env, init)
        env.run(env.T); res.append((env.orders(), env.counters().copy()))
        env.reset_again(); check_lists(env, init)                  # (the image, where the form applies)
        env.run(env.T); res.append((env.orders(), env.counters().copy()))
        env.reset_again(); env.run(env.T // 2)
        env.reset(init_b); check_lists(env, init_b)                # new start nodes: sorted again, a new image
        env.reset_again(); check_lists(env, init_b)
        env.run(env.T); res.append((env.orders(), env.counters().copy()))
        env.set_idle_cap(env.idle_cap + 64)
        env.reset_again(); check_lists(env, init_b)                # other idle tables: not the old image
        env.run(env.T); res.append((env.orders(), env.counters().copy()))
        outs.append(res)
        env.close()
    for a, b in zip(outs[0], outs[1]):
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(a[0][k], b[0][k])
        np.testing.assert_array_equal(a[1], b[1])
    for res in outs:                                               # same nodes, same day: same results
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(res[0][0][k], res[1][0][k])
            np.testing.assert_array_equal(res[2][0][k], res[3][0][k])


@pytest.mark.parametrize("shape", [(300, 12, 150, 5), (4139, 192, 3000, 4), (4096, 40, 700, 3), (1, 1, 40, 2), (257, 9, 1, 3)])
def test_reset_random_draws_the_reference_start_nodes_on_the_device(shape):
    """vds_reset_random: InitVehiclesIntoCluster (:249-258) with `random.Random(seed_r)` per replica, drawn by a device MT19937 -
    vehicle for vehicle the nodes CPython draws (random.Random(seed).choice(range(N)), retried while the node is in no cluster),
    for seeds of one and two 32-bit words, N at and next to a power of two, a city with nodes outside every cluster; then the
    day from those nodes equals the day from the same nodes uploaded by vds_reset."""
    import random as pyrandom
    from vehicles_dispatch_simulator_amd import workloads
    N, Cn, V, R = shape
    w = workloads.tiny(N=N, C=Cn, vehicles=V, orders=800, seed=3 + N)
    n2c = np.asarray(w.city.node2cluster).copy()
    if N > 64:
        n2c[5:N:7] = -1                                   # nodes in no cluster: those draws are retried (:254)
    seeds = np.array([0, 1, 2**32 - 1, 2**32, 2**63 + 12345][:R], dtype=np.uint64)
    env = BatchedDispatchEnv(w.city.cost, n2c, w.nbr_off, w.nbr_idx, replicas=R, vehicles=V, depth_limit=0, neighbor_can_server=False)
    keep = (n2c[np.asarray(w.pickup)] >= 0) & (n2c[np.asarray(w.delivery)] >= 0)
    env.load_orders(np.asarray(w.release_min)[keep], np.asarray(w.pickup)[keep], np.asarray(w.delivery)[keep])
    env.reset_random(seeds)
    exp = np.zeros((R, V), dtype=np.int32)
    for r in range(R):
        rng = pyrandom.Random(int(seeds[r]))
        for v in range(V):
            while True:
                node = rng.choice(range(N))
                if n2c[node] >= 0:
                    break
            exp[r, v] = node
        got = env.vehicles(r)
        assert (got["state"] == 0).all()
        np.testing.assert_array_equal(got["node"], exp[r], err_msg="replica %d" % r)
    env.run(env.T)
    a = (env.orders(), env.counters().copy())
    env.reset_again()                                     # the drawn nodes stay resident
    env.run(env.T)
    b = (env.orders(), env.counters().copy())
    env.reset(exp)
    env.run(env.T)
    c = (env.orders(), env.counters().copy())
    env.close()
    for other in (b, c):
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(a[0][k], other[0][k])
        np.testing.assert_array_equal(a[1], other[1])


@pytest.mark.parametrize("host_checks", ["0", "1"])
def test_reset_checks_start_nodes_on_the_device_and_keeps_the_old_ones_on_failure(host_checks, monkeypatch):
    """vds_reset validates the uploaded start nodes with one kernel (VDS_RESET_HOST_CHECKS=1: the host loops of rounds 1-3): the FIRST
    vehicle on a node outside every cluster is named (:254), a failed reset leaves the previous episode's start nodes in place
    (vds_reset_again replays them), the idle capacity follows the fullest start list as before."""
    from vehicles_dispatch_simulator_amd import workloads
    monkeypatch.setenv("VDS_RESET_HOST_CHECKS", host_checks)
    w = workloads.tiny(N=300, C=12, vehicles=200, orders=1200, seed=41)
    n2c = np.asarray(w.city.node2cluster).copy()
    n2c[7] = -1
    keep = (n2c[np.asarray(w.pickup)] >= 0) & (n2c[np.asarray(w.delivery)] >= 0)
    R, V = 3, 200
    env = BatchedDispatchEnv(w.city.cost, n2c, w.nbr_off, w.nbr_idx, replicas=R, vehicles=V, depth_limit=0, neighbor_can_server=False)
    env.load_orders(np.asarray(w.release_min)[keep], np.asarray(w.pickup)[keep], np.asarray(w.delivery)[keep])
    rng = np.random.default_rng(5)
    valid = np.flatnonzero(n2c >= 0)
    good = valid[rng.integers(0, valid.size, size=(R, V))].astype(np.int32)
    env.reset(good)
    env.run(env.T)
    ref = (env.orders(), env.counters().copy())
    for bad_value in (7, -3, 300, 10**6):
        bad = good.copy()
        bad[1, 57] = bad_value
        bad[2, 3] = 7                                        # a later offender: the first one is reported
        with pytest.raises(Exception, match=r"vehicle 257 starts on node %d which is in no cluster" % bad_value):
            env.reset(bad)
    env.reset_again()                                        # the nodes of the last successful reset
    env.run(env.T)
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(env.orders()[k], ref[0][k])
    np.testing.assert_array_equal(env.counters(), ref[1])
    crowded = np.full((R, V), valid[0], np.int32)            # every vehicle in one cluster: the capacity grows to hold them
    env.reset(crowded)
    assert env.idle_cap >= V
    assert env.lists(2)["idle_off"][int(n2c[valid[0]]) + 1] - env.lists(2)["idle_off"][int(n2c[valid[0]])] == V
    env.close()
