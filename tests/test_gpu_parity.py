"""Parity of the HIP tick path (through the C ABI) against the CPU oracle and the golden vectors
captured from the unmodified reference.  Integer bookkeeping: bit-exact, per tick:
counters, per-cluster observations, idle lists in list order, arrival dicts in insertion order,
and per-order (status, vehicle, wait) at the end of the day."""
import random

import numpy as np
import pytest

from helpers import dispatch_by_tick, engine_settings, golden_names, load_golden, make_oracle
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv, synth
from vehicles_dispatch_simulator_amd.env import neighbors_to_csr

pytestmark = pytest.mark.gpu

TINY = golden_names("tiny_")


def make_env(g, R, **kw):
    environ = kw.pop("environ", None)
    if environ:       # library switches read when the order tables are loaded (e.g. VDS_DENSE_DFS=0: neighbour search on the wide layout)
        import os
        saved = {k: os.environ.get(k) for k in environ}
        os.environ.update(environ)
        try:
            return make_env(g, R, **kw)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=int(g["V"]),
                             depth_limit=int(g["depth_limit"]), neighbor_can_server=bool(g["neighbor_can_server"]),
                             **{**engine_settings(g), **kw})
    env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
    return env


def check_lists(env, r, o, t):
    L, G = o.lists(), env.lists(r)
    np.testing.assert_array_equal(G["idle_off"], L["idle_off"], err_msg="tick %d replica %d idle_off" % (t, r))
    np.testing.assert_array_equal(G["idle_veh"], L["idle_veh"], err_msg="tick %d replica %d idle order" % (t, r))
    np.testing.assert_array_equal(G["arr_off"], L["arr_off"], err_msg="tick %d replica %d arr_off" % (t, r))
    np.testing.assert_array_equal(G["arr_veh"], L["arr_veh"], err_msg="tick %d replica %d arrival dict order" % (t, r))
    np.testing.assert_array_equal(G["arr_min"], L["arr_min"])
    veh = o.vehicles()
    n = L["idle_off"][-1]
    np.testing.assert_array_equal(G["idle_node"][:n], veh["loc"][L["idle_veh"][:n]])
    na = L["arr_off"][-1]
    np.testing.assert_array_equal(G["arr_node"][:na], veh["dest"][L["arr_veh"][:na]])
    np.testing.assert_array_equal(G["arr_order"][:na], veh["order"][L["arr_veh"][:na]])
    # per-vehicle view (vds_read_vehicles) against the oracle's Vehicle fields
    W = env.vehicles(r)
    idle = np.zeros(env.V, dtype=bool); idle[L["idle_veh"][:n]] = True
    fly = L["arr_veh"][:na]
    assert ((W["state"] == 0) == idle).all() and set(np.flatnonzero(W["state"] > 0)) == set(fly.tolist())
    np.testing.assert_array_equal(W["node"][idle], veh["loc"][idle])
    np.testing.assert_array_equal(W["node"][fly], veh["dest"][fly])
    np.testing.assert_array_equal(W["order"][fly], veh["order"][fly])
    np.testing.assert_array_equal(W["state"][fly], np.where(veh["order"][fly] >= 0, 1, 2))
    np.testing.assert_array_equal(W["arrive_min"][fly], L["arr_min"][:na])
    assert (W["arrive_min"][idle] == -1).all() and (W["order"][idle] == -1).all()
    cl_idle = np.empty(env.V, dtype=np.int64); cl_idle[L["idle_veh"][:n]] = np.repeat(np.arange(env.C), np.diff(L["idle_off"]))
    np.testing.assert_array_equal(W["cluster"][idle], cl_idle[idle])


def run_day(g, R, same_init, list_every=1, device_dispatch=False, **kw):
    V, N = int(g["V"]), int(g["N"])
    if device_dispatch:
        import torch
        kw = dict(kw, stream=torch.cuda.current_stream().cuda_stream)
    valid = g["node2cluster"] >= 0
    init = np.empty((R, V), dtype=np.int32)
    init[0] = g["veh_node"]
    for r in range(1, R):
        init[r] = g["veh_node"] if same_init else synth.init_vehicle_nodes(random.Random(1000 + r), N, V, valid)
    # SupplyExpect kept in place (vds_supply_inplace: dense layout, switched on by the first request - before the reset): in the
    # modes that say so and in every dense-tick variant; the other runs take their `supply` plane from the arrival tables
    want_sup = kw.pop("supply_inplace", "dense_debug" in kw)
    env = make_env(g, R, **kw)
    sup_ring = None
    if want_sup:
        try:
            sup_ring, sup_slot = env.supply_inplace_torch()
        except Exception as e:      # (the library keeps the wide layout for this fixture / mode)
            assert "dense layout only" in str(e), e
    env.reset(init)
    oracles = []
    for r in range(R):
        o = make_oracle(g)
        o.reset(init[r])
        oracles.append(o)
    T = env.T
    assert T == int(g["n_ticks"]) == oracles[0].num_ticks
    disp = dispatch_by_tick(g)
    extra = int(g["dispatch_extra_minutes"]) if "dispatch_extra_minutes" in g else 0
    for t in range(T):
        env.step()
        for o in oracles:
            o.begin_tick()
        ob = env.obs()
        cn = env.counters()
        if sup_ring is not None:
            np.testing.assert_array_equal(sup_ring[int(sup_slot.item())].cpu().numpy(), ob["supply"], err_msg="tick %d supply in place" % t)
        for r, o in enumerate(oracles):
            oo, oc = o.obs(), o.counters()
            for a, b in (("idle_pre", "idle_pre"), ("idle_now", "idle_post"), ("supply", "supply"), ("cl_orders", "cl_orders"), ("inflight", "inflight")):
                np.testing.assert_array_equal(ob[a][r], oo[b], err_msg="tick %d replica %d obs %s" % (t, r, a))
            for i, k in enumerate(("order_num", "reject_num", "matched", "wait_sum", "dispatch_num", "dispatch_cost")):
                assert cn[r, i] == oc[k], (t, r, k, cn[r, i], oc[k])
            assert cn[r, 7] == oc["evals"], (t, r, "evals")
        # replica 0 also against the reference's own per-tick record
        np.testing.assert_array_equal(ob["supply"][0], g["t_supply"][t])
        np.testing.assert_array_equal(ob["idle_now"][0], g["t_idle_post"][t])
        assert cn[0, 0] == g["t_order_num"][t] and cn[0, 1] == g["t_reject_num"][t] and cn[0, 3] == g["t_wait_sum"][t]
        if t % list_every == 0 or t in disp:
            for r, o in enumerate(oracles):
                check_lists(env, r, o, t)
        if t in disp:
            rows = np.array(disp[t])
            L = oracles[0].lists()
            pos = []
            for veh, cl in zip(rows[:, 1], rows[:, 2]):
                seg = L["idle_veh"][L["idle_off"][cl]:L["idle_off"][cl + 1]]
                pos.append(int(np.flatnonzero(seg == veh)[0]))
            nrep = R if same_init else 1
            rep = np.repeat(np.arange(nrep), len(rows))
            if device_dispatch:       # the same hook body as a device-resident [R, K, 3] action tensor
                acts = np.full((R, len(rows) + 3, 3), -1, dtype=np.int32)
                acts[:nrep, 1:len(rows) + 1, 0] = rows[:, 2]; acts[:nrep, 1:len(rows) + 1, 1] = pos; acts[:nrep, 1:len(rows) + 1, 2] = rows[:, 4]
                env.apply_dispatch_torch(torch.from_numpy(acts).cuda())
            elif extra:       # the hook body wrote its own arrival time (golden tiny_dispatch_delay): vds_apply_dispatch_ex
                env.apply_dispatch(rep, np.tile(rows[:, 2], nrep), np.tile(pos, nrep), np.tile(rows[:, 4], nrep),
                                   arrive_min=np.tile(oracles[0].now_min + rows[:, 5] + extra, nrep))
            else:
                env.apply_dispatch(rep, np.tile(rows[:, 2], nrep), np.tile(pos, nrep), np.tile(rows[:, 4], nrep))
            for o in oracles[:nrep]:
                if extra:
                    o.dispatch_at(rows[:, 1], rows[:, 4], arrive_min=o.now_min + rows[:, 5] + extra)
                else:
                    o.dispatch(rows[:, 1], rows[:, 4])
            for r, o in enumerate(oracles):
                check_lists(env, r, o, t)
            np.testing.assert_array_equal(env.obs()["idle_now"][0], g["t_idle_after_dispatch"][t])
        env.advance()
        for o in oracles:
            o.end_tick()
    env.sync()
    od = env.orders()
    for r, o in enumerate(oracles):
        oo = o.orders()
        np.testing.assert_array_equal(od["status"][r], oo["status"])
        np.testing.assert_array_equal(od["vehicle"][r], oo["vehicle"])
        np.testing.assert_array_equal(od["wait"][r], oo["wait"])
    np.testing.assert_array_equal(od["status"][0], g["o_status"])
    np.testing.assert_array_equal(od["vehicle"][0], g["o_vehicle"])
    np.testing.assert_array_equal(od["wait"][0], g["o_wait"])
    cn = env.counters()
    for r, o in enumerate(oracles):
        oc = o.counters()
        assert cn[r, 6] == oc["sum_order_value"]
    assert cn[0, 6] == int(g["sum_order_value"]) and cn[0, 0] == int(g["order_num"]) and cn[0, 1] == int(g["reject_num"])
    assert cn[0, 4] == int(g["dispatch_num"]) and cn[0, 5] == int(g["dispatch_cost"])
    tot = env.total_counters()
    np.testing.assert_array_equal(tot[:6], cn[:, :6].sum(axis=0))
    env.close()


MODES = {
    "fast": {},                              # the default kernels
    "fast_sup": {"supply_inplace": True},    # ... with SupplyExpect kept in place (per-arrival-slot counters bumped at match time)
    "generic": {"force_generic": True},      # one wavefront per bucket
    "far": {"ring_ticks": 2},                # nearly every trip outlives the ring: far tables + migration
    "far_generic": {"ring_ticks": 4, "force_generic": True},
    "dfs_v2": {"force_generic": 3},          # neighbour search by lower-bound rounds (k_tick_replica2: what the hybrid tick falls back to)
    # neighbour search on the WIDE layout (k_tick_rows in stamp mode + the committing k_dfs_walk of rounds 2-5): what the library keeps for
    # order days per replica, costs beyond a byte, orders without a static arrival slot; the default since round 6 is the dense layout
    # (k_tick_dense in stamp form + the walk writing only what moved: "fast" on a searching fixture)
    "dfs_wide": {"environ": {"VDS_DENSE_DFS": "0"}},
    # dense tick (k_tick_dense: the default without neighbour search; 8 lanes per replica by default) - 16 lanes per replica; tiny fast-path tables
    # (buckets that outgrow them take dense_bucket_slow); slow path only; far tables.  With neighbour search or a live pickup
    # window the library keeps the wide layout (these fixtures then repeat the default run).
    "dense16": {"dense_debug": (16, 0, 0, 0)},
    "dense_tiny": {"dense_debug": (16, 8, 2, 0)},
    "dense_tiny8": {"dense_debug": (8, 12, 3, 0)},
    "dense_tiny20": {"dense_debug": (8, 20, 1, 0)},
    "dense_slow": {"dense_debug": (16, 0, 0, 1)},
    "dense_far": {"dense_debug": (16, 0, 0, 0), "ring_ticks": 2},
    # the same with the arrival ring instead of static arrival slots (what the dense tick does when arrivals can lie more than
    # DENSE_PULL_WMAX slots behind their earliest slot): force_slow bit 1
    "dense_ring": {"dense_debug": (16, 0, 0, 2)},
    "dense_ring8_tiny": {"dense_debug": (8, 12, 3, 2)},
    "dense_ring_slow": {"dense_debug": (16, 0, 0, 3)},
    "dense_ring_far": {"dense_debug": (8, 0, 0, 2), "ring_ticks": 2},
    # the dense tick's two forms (8 lanes per replica / 16 lanes with 256-entry tables) alternating slot by slot: what the per-slot
    # choice of adapt_dense mixes within a day
    "dense_alt": {"environ": {"VDS_DENSE_TICK_FORMS": "alt"}},
    "dense_alt_tiny8": {"dense_debug": (8, 12, 3, 0), "environ": {"VDS_DENSE_TICK_FORMS": "alt"}},
    "rows": {"force_generic": 5},            # wide layout + the row-mapped kernel where the dense tick is the default
    "rows_far": {"force_generic": 5, "ring_ticks": 2},
    # a ring horizon beyond the dense keys' 32 slots (any power of two is valid): the library keeps the wide layout and must not
    # leave the dense tick's static-arrival-slot flag behind - observations and arrival lists read D.arr only when the dense tick
    # wrote it
    "ring64": {"ring_ticks": 64},
}
DENSE = [m for m in MODES if m.startswith("dense")]
# every mode on a representative handful of fixtures; the default kernels, the generic kernels and the far path on all of them
# (the full product was 17 modes x 19 fixtures: most of the GPU suite's run time for pairs that add no new path)
ALL_MODES_ON = ("tiny_kmeans", "tiny_grid", "tiny_dispatch", "tiny_sort_ties", "tiny_fraccost", "tiny_kmeans_dfs2", "tiny_dispatch_dfs2", "tiny_window4_dfs2")
BASE_MODES = ("fast", "fast_sup", "generic", "far", "dfs_v2", "dfs_wide", "dense16", "ring64")


def _applies(name, mode, every_pair=False):
    """Mode x fixture pairs that would only repeat the default run are left out: the dense-tick variants (and the explicit
    row-mapped kernel) on fixtures with neighbour search or a live pickup window (the library keeps its own choice there),
    the fallback neighbour-search kernel on fixtures without neighbour search.  every_pair: also the pairs the parametrised test
    leaves to test_remaining_mode_fixture_pairs_in_worker_processes."""
    g = load_golden(name)
    searching = bool(g["neighbor_can_server"]) and int(g["depth_limit"]) > 0
    if name not in ALL_MODES_ON and mode not in BASE_MODES and not every_pair:
        return False
    if mode.startswith("dense_ring") or mode.startswith("rows"):
        return not searching and "window" not in name
    if mode.startswith("dense"):       # (neighbour search runs on the dense layout too: stamp form; a live pickup window keeps the wide layout)
        return "window" not in name
    if mode.startswith("dfs_"):
        return searching
    return True


# in this process: the default kernels (without / with SupplyExpect in place) and the generic kernels on every fixture; every other
# applicable (fixture, mode) pair - the whole product - runs in test_mode_fixture_product_in_worker_processes
IN_PROCESS_MODES = ("fast", "fast_sup", "generic")
RAGGED = [(n, m) for n in ["tiny_kmeans", "tiny_kmeans_dfs2", "tiny_grid"] for m in ["fast", "generic", "rows"] + DENSE
          if _applies(n, m) and (n != "tiny_grid" or m in ("fast", "dense16", "dense_tiny", "dense_slow"))]
RAGGED_IN_PROCESS = ("fast", "dense16")


@pytest.mark.parametrize("name,mode", [(n, m) for n in TINY for m in IN_PROCESS_MODES if _applies(n, m, every_pair=True)])
def test_tiny_golden_per_tick(name, mode):
    g = load_golden(name)
    run_day(g, R=3, same_init=bool(len(g["dispatch_log"])), **MODES[mode])


PRODUCT_NPROC = 8          # (beside the other tests on a 16-core box: more of them slow the pytest process itself down)


def product_pairs():
    pairs = [(n, m, "tick") for n in TINY for m in MODES if _applies(n, m, every_pair=True) and m not in IN_PROCESS_MODES]
    pairs += [(n, m, "ragged") for n, m in RAGGED if m not in RAGGED_IN_PROCESS]
    pairs.sort(key=lambda x: x[2] != "ragged")        # (the long items first)
    return pairs


def product_worker_argvs():
    import json, os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    pairs = product_pairs()
    return [[sys.executable, os.path.join(here, "parity_worker.py"), json.dumps(pairs[i::PRODUCT_NPROC])] for i in range(PRODUCT_NPROC)]


def test_mode_fixture_product_in_worker_processes():
    """The rest of the mode x fixture product - every engine mode on every tiny fixture it applies to, per tick incl. container order
    (run_day, as test_tiny_golden_per_tick) - and the ragged 37-replica days of the modes test_many_replicas_ragged leaves out, dealt
    to eight worker processes (tests/parity_worker.py): a day's cost is host work (oracle steps, container comparisons per tick).
    The workers are started when the session's collection is final (tests/conftest.py) and run beside the tests before this one."""
    import conftest
    pairs = product_pairs()
    conftest.start_workers("parity_product", product_worker_argvs())
    done = 0
    for i, (rc, out) in enumerate(conftest.collect_workers("parity_product")):
        assert rc == 0 and "WORKER DONE" in out, "worker %d:\n%s" % (i, out[-3000:])
        done += int(out.split("WORKER DONE")[1].split()[0])
    assert done == len(pairs) and len(pairs) > 200


@pytest.mark.parametrize("name", ["tiny_dispatch", "tiny_dispatch_dfs2"])
def test_device_resident_dispatch_tensor(name):
    """vds_apply_dispatch_device: the reference-side dispatch log replayed as a GPU action tensor (with empty slots)."""
    g = load_golden(name)
    run_day(g, R=5, same_init=True, device_dispatch=True)


@pytest.mark.parametrize("name,mode", [(n, m) for n, m in RAGGED if m in RAGGED_IN_PROCESS])
def test_many_replicas_ragged(name, mode):
    """R not a multiple of the workgroup's replica run; every replica its own vehicle seed."""
    g = load_golden(name)
    run_day(g, R=37, same_init=False, list_every=29, **MODES[mode])


def _burst(g, n_orders):
    """n identical orders in one minute, all vehicles parked on the pickup node: every match ends
    its trip in the same cluster at the same tick (maximum pressure on one ring slot)."""
    g = dict(g)
    P = int(np.flatnonzero(g["node2cluster"] == 2)[0])
    X = int(np.flatnonzero(g["node2cluster"] == 7)[0])
    g["o_release_min"] = np.zeros(n_orders + 1, dtype=np.int32)
    g["o_pickup"] = np.full(n_orders + 1, P, dtype=np.int32)
    g["o_delivery"] = np.full(n_orders + 1, X, dtype=np.int32)
    g["V"] = np.int64(120)
    return g, P


@pytest.mark.parametrize("mode", ["dense_ring", "rows"])
def test_tight_ring_cap_overflow_is_reported(mode):
    """(With static arrival slots - the dense tick's default - order-carrying arrivals never touch the ring: nothing can overflow,
    the same day simply runs, see test_burst_of_identical_orders.)"""
    g, P = _burst(load_golden("tiny_kmeans"), 100)
    env = make_env(g, 2, ring_cap=16, idle_cap=128, **MODES[mode])
    env.reset(np.full((2, 120), P, dtype=np.int32))
    with pytest.raises(Exception, match="arrival ring overflow"):
        env.run(env.T)
        env.sync()
    env.close()


@pytest.mark.parametrize("mode", ["fast", "generic", "far"] + DENSE)
def test_burst_of_identical_orders(mode):
    """100 identical orders / 120 co-located vehicles: ties everywhere, 100 arrivals in one slot."""
    g, P = _burst(load_golden("tiny_kmeans"), 100)
    env = make_env(g, 2, ring_cap=128, idle_cap=128, **MODES[mode])
    init = np.full((2, 120), P, dtype=np.int32)
    env.reset(init)
    o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], 0, False, g["o_release_min"], g["o_pickup"], g["o_delivery"], 120)
    o.reset(init[0])
    for t in range(env.T):
        env.step(); o.begin_tick()
        check_lists(env, 1, o, t)
        np.testing.assert_array_equal(env.obs()["supply"][1], o.obs()["supply"])
        env.advance(); o.end_tick()
    od, oo = env.orders(), o.orders()
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(od[k][0], oo[k])
    env.close()


def test_small_caps_overflow_is_reported():
    g = load_golden("tiny_kmeans")
    env = make_env(g, 2, idle_cap=64)
    init = np.tile(np.full(int(g["V"]), int(np.flatnonzero(g["node2cluster"] == 0)[0]), dtype=np.int32), (2, 1))
    with pytest.raises(Exception, match="idle table overflow"):
        env.reset(init)   # 150 vehicles in one cluster > idle_cap 64
    env.close()


@pytest.mark.parametrize("mode", ["fast", "generic"] + DENSE)
def test_all_vehicles_in_one_cluster_oversize_bucket(mode):
    """> 256 idle vehicles in one cluster exercises the deferred 16-slot kernel."""
    g = load_golden("tiny_kmeans")
    V = 600
    g = dict(g)
    g["V"] = np.int64(V)
    node = int(np.flatnonzero(g["node2cluster"] == 5)[0])
    nodes5 = np.flatnonzero(g["node2cluster"] == 5)
    init = nodes5[np.arange(V) % nodes5.size].astype(np.int32)[None, :].repeat(2, axis=0)
    env = make_env(g, 2, idle_cap=640, **MODES[mode])
    env.reset(init)
    o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], int(g["depth_limit"]), False,
               g["o_release_min"], g["o_pickup"], g["o_delivery"], V)
    o.reset(init[0])
    for t in range(env.T):
        env.step(); o.begin_tick()
        if t % 10 == 0:
            check_lists(env, 1, o, t)
        env.advance(); o.end_tick()
    od, oo = env.orders(), o.orders()
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(od[k][1], oo[k])
    assert node >= 0
    env.close()


def test_zero_copy_torch_observations_match_host_read():
    import torch
    g = load_golden("tiny_kmeans")
    env = make_env(g, 4, stream=torch.cuda.current_stream().cuda_stream)
    env.reset(np.tile(g["veh_node"], (4, 1)))
    for _ in range(30):
        env.step(); env.advance()
    host = env.obs()
    dev = env.obs_torch()
    torch.cuda.synchronize()
    assert dev.shape == (5, 4, int(g["C"])) and dev.dtype == torch.int32 and dev.is_cuda
    for i, k in enumerate(("idle_pre", "idle_now", "supply", "cl_orders", "inflight")):
        np.testing.assert_array_equal(dev[i].cpu().numpy(), host[k])
    # a subset of the planes (vds_obs_device_planes): the other planes keep what they held, the wanted ones are exact per slot
    for t in range(10):
        env.step()
        host = env.obs()                            # (all five planes)
        stale = env.obs_torch().clone()
        stale[4] = -7
        env.obs_torch()[4] = -7
        part = env.obs_torch(inflight=False)
        torch.cuda.synchronize()
        for i, k in enumerate(("idle_pre", "idle_now", "supply", "cl_orders")):
            np.testing.assert_array_equal(part[i].cpu().numpy(), host[k], err_msg="%d %s" % (t, k))
        assert (part[4] == -7).all()
        env.advance()
    with pytest.raises(Exception, match="planes"):
        env.obs_device_ptr(0)
    # per-replica counters, zero copy (device order: orders, rejects, wait, matched value, evals, arrivals, dispatch, cost)
    cn = env.counters()
    ct = env.counters_torch()
    torch.cuda.synchronize()
    assert ct.shape == (4, 8) and ct.dtype == torch.int64 and ct.is_cuda
    ct = ct.cpu().numpy()
    np.testing.assert_array_equal(ct[:, 0], cn[:, 0]); np.testing.assert_array_equal(ct[:, 1], cn[:, 1])
    np.testing.assert_array_equal(ct[:, 2], cn[:, 3]); np.testing.assert_array_equal(ct[:, 4], cn[:, 7])
    np.testing.assert_array_equal(ct[:, 0] - ct[:, 1], cn[:, 2])
    np.testing.assert_array_equal(ct[:, 6], cn[:, 4]); np.testing.assert_array_equal(ct[:, 7], cn[:, 5])
    env.close()


@pytest.mark.parametrize("name", ["tiny_fraccost", "tiny_fraccost_dfs2"])
def test_fractional_road_costs_float_array_and_data_dir(name, tmp_path):
    """The float boundary (north_star: "within 1e-6 on float distances"): the only floating point on the path is ``int(float)``
    in ``RoadCost`` (simulator.py:263-264).  The fixtures were captured from the unmodified reference fed an ``AccurateMap.csv``
    in FRACTIONAL minutes (9.9 / 10.0 / 0.4, asymmetric pairs).  (a) the float matrix handed to ``BatchedDispatchEnv`` as is,
    (b) the same table read from a reference-style data directory by ``world.load_world`` (neighbour table computed by the HIP
    reduction - no cache file).  Once truncated the path is integer: results are bit-exact, tolerance 0."""
    import os
    import time
    from oracle.ref_harness import write_reference_data_dir
    from vehicles_dispatch_simulator_amd import world
    g = load_golden(name)
    F = g["cost_float"]
    assert F.dtype == np.float64 and (np.trunc(F) != np.rint(F)).any() and (np.trunc(F) == g["cost"]).all()

    def day(cost, node2cluster, nbr_off, nbr_idx, depth, rel, pk, dl):
        env = BatchedDispatchEnv(cost, node2cluster, nbr_off, nbr_idx, replicas=2, vehicles=int(g["V"]), depth_limit=depth,
                                 neighbor_can_server=bool(g["neighbor_can_server"]), **engine_settings(g))
        env.load_orders(rel, pk, dl)
        env.reset(np.tile(g["veh_node"], (2, 1)))
        for t in range(env.T):
            env.step()
            cn = env.counters()
            assert cn[1, 0] == g["t_order_num"][t] and cn[1, 1] == g["t_reject_num"][t] and cn[1, 3] == g["t_wait_sum"][t], t
            np.testing.assert_array_equal(env.obs()["supply"][1], g["t_supply"][t])
            env.advance()
        od, cn = env.orders(), env.counters()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(od[k][1], g["o_" + k], err_msg=k)
        assert cn[1, 6] == int(g["sum_order_value"]) and cn[1, 3] == int(g["wait_sum"])
        env.close()

    day(F, g["node2cluster"], g["nbr_off"], g["nbr_idx"], int(g["depth_limit"]), g["o_release_min"], g["o_pickup"], g["o_delivery"])

    city = synth.make_city(int(g["city_seed"]), N=int(g["N"]), C=int(g["C"]), frac=True)
    start, pick, dele = synth.make_orders(int(g["order_seed"]), city.N, int(g["n_orders_raw"]))
    os.environ["TZ"] = "UTC"
    time.tzset()
    write_reference_data_dir(str(tmp_path), city, start, pick, dele, n_drivers=int(g["V"]), cluster_mode=str(g["cluster_mode"]))
    W = world.load_world(os.path.join(str(tmp_path), "data"), cluster_mode=str(g["cluster_mode"]), local_region_bound=synth.DEFAULT_BOUND,
                         side_length_meter=float(g["side_m"]), vehicles_service_meter=float(g["service_m"]))
    np.testing.assert_array_equal(W.cost, g["cost"])
    off, idx = neighbors_to_csr(W.neighbors)
    np.testing.assert_array_equal(idx, g["nbr_idx"])
    day(W.cost, W.node2cluster, off, idx, W.depth_limit, W.o_release_min, W.o_pickup, W.o_delivery)
