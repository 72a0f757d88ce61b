"""Randomised differential test: many small random cities / configurations, HIP path (through the C ABI) vs the
CPU oracle, bit-exact on per-order results, counters and the final container state.  Seeds are fixed."""
import os
import random

import numpy as np
import pytest

from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv, neighbors_to_csr, synth

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(20, 220))
    C = int(rng.integers(1, min(24, N)))
    city = synth.make_city(seed * 7 + 1, N=N, C=C, with_neighbors=False)
    cost = city.cost.copy()
    style = int(rng.integers(0, 4))
    if style == 1:                                   # many ties
        cost = (cost // 7).astype(np.int32)
    elif style == 2:                                 # long trips (far tables)
        cost = (cost * int(rng.integers(3, 9))).astype(np.int32)
    elif style == 3:                                 # negative entries -> generic kernels
        cost = (cost - int(rng.integers(1, 6))).astype(np.int32)
    n2c = city.node2cluster.copy()
    if rng.random() < 0.3:                           # some nodes outside every cluster
        out = rng.random(N) < 0.15
        if (~out).sum() >= 4:
            n2c[out] = -1
    nbr = []
    for c in range(C):
        k = int(rng.integers(0, min(C, 6)))
        nbr.append(rng.choice(C, size=k, replace=False).tolist() if k else [])
    V = int(rng.integers(0, 140))
    O = int(rng.integers(2, 1400))
    valid_nodes = np.flatnonzero(n2c >= 0)
    span = int(rng.integers(30, 1440))
    rel = np.sort(rng.integers(0, span, size=O)).astype(np.int32)
    if rng.random() < 0.25:                          # a few out-of-order releases
        idx = rng.choice(O, size=max(1, O // 20), replace=False)
        rel[idx] = rng.integers(0, span, size=idx.size)
        rel[0] = min(int(rel[0]), int(rel.min()))
    pick = rng.choice(valid_nodes, size=O).astype(np.int32)
    dele = rng.choice(valid_nodes, size=O).astype(np.int32)
    cfg = dict(neighbor=bool(rng.random() < 0.5), depth=int(rng.integers(-1, 4)),
               threshold=int(rng.choice([600_000_000_000, 600_000_000_000, 40, 8, 0])),
               ring_ticks=int(rng.choice([32, 32, 8, 2])), force_generic=int(rng.choice([0, 0, 0, 1, 5, 3, 0])),
               tick=int(rng.choice([10, 10, 5, 15])), R=int(rng.integers(1, 9)), dispatch=bool(rng.random() < 0.4))
    return cost, n2c, nbr, V, rel, pick, dele, valid_nodes, cfg


def medium_case(seed):
    """Mid-size cities: lists longer than the register tables, bursts of arrivals, several worklist steps per round."""
    rng = np.random.default_rng(50_000 + seed)
    N = int(rng.integers(300, 900))
    C = int(rng.integers(20, 150))
    city = synth.make_city(seed * 11 + 5, N=N, C=C, with_neighbors=False)
    cost = city.cost.copy()
    if rng.random() < 0.3:
        cost = (cost // 5).astype(np.int32)          # many ties
    n2c = city.node2cluster.copy()
    nbr = []
    for c in range(C):
        k = int(rng.integers(0, 9))
        nbr.append(rng.choice(C, size=min(k, C), replace=False).tolist() if k else [])
    V = int(rng.integers(150, 1000))
    O = int(rng.integers(2000, 12000))
    valid_nodes = np.flatnonzero(n2c >= 0)
    rel = np.sort(rng.integers(0, 1440, size=O)).astype(np.int32)
    hot = rng.random() < 0.5                          # half of the days: deliveries concentrate on a few clusters
    pick = rng.choice(valid_nodes, size=O).astype(np.int32)
    dele = rng.choice(valid_nodes, size=O).astype(np.int32)
    if hot:
        hot_nodes = np.flatnonzero(n2c < max(1, C // 10))
        sel = rng.random(O) < 0.7
        dele[sel] = rng.choice(hot_nodes, size=int(sel.sum()))
    cfg = dict(neighbor=bool(rng.random() < 0.8), depth=int(rng.integers(0, 5)),
               threshold=int(rng.choice([600_000_000_000, 600_000_000_000, 30])),
               ring_ticks=int(rng.choice([32, 32, 4])), force_generic=int(rng.choice([0, 0, 0, 5, 3, 0])),
               tick=int(rng.choice([10, 10, 5])), R=int(rng.integers(1, 4)), dispatch=bool(rng.random() < 0.3))
    return cost, n2c, nbr, V, rel, pick, dele, valid_nodes, cfg


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VDS_FUZZ_MEDIUM_N", "12")))))
def test_medium_city_matches_oracle(seed):
    run_case(seed, medium_case(seed), idle_cap=1024)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("VDS_FUZZ_N", "48")))))
def test_random_city_matches_oracle(seed):
    run_case(seed, random_case(seed))


def run_case(seed, case, idle_cap=None, whole_day=False):
    """whole_day: no hooks - the day as one vds_run (the day graph, replica groups as branches), every replica's final results,
    counters, observations and containers against the oracle's; the volume form (test_fuzz_volume_in_worker_processes)."""
    cost, n2c, nbr, V, rel, pick, dele, valid_nodes, cfg = case
    if whole_day:
        cfg = dict(cfg, dispatch=False)
    off, idx = neighbors_to_csr(nbr)
    R, N = cfg["R"], cost.shape[0]
    rng = np.random.default_rng(1000 + seed)
    init = rng.choice(valid_nodes, size=(R, V)).astype(np.int32) if V else np.zeros((R, 0), np.int32)
    # the cases that run the default kernels take the dense tick (k_tick_dense) when its preconditions hold: random lanes per
    # replica / fast-path table sizes / forced slow path (its own random stream: the cases above stay what they were); with
    # neighbour search or a live pickup window the library keeps the wide layout
    lr = np.random.default_rng(90_000 + seed)
    fg, kw = cfg["force_generic"], {}
    if fg == 0:
        kw["dense_debug"] = (int(lr.choice([16, 8])), int(lr.choice([0, 0, 8, 24, 40])), int(lr.choice([0, 0, 2, 5])), int(lr.random() < 0.1) | (2 if lr.random() < 0.3 else 0))
    # a third of the cases with neighbour search on the default kernels keep the WIDE layout (VDS_DENSE_DFS=0, read when the static tables
    # are loaded: k_tick_rows in stamp mode + the committing walk) instead of the dense layout's stamp form
    if fg == 0 and cfg["neighbor"] and lr.random() < 0.34:
        os.environ["VDS_DENSE_DFS"] = "0"
    # ... and a third of the dense-tick cases alternate its two forms slot by slot (what the per-slot choice mixes within a day)
    if fg == 0 and np.random.default_rng(170_000 + seed).random() < 0.34:
        os.environ["VDS_DENSE_TICK_FORMS"] = "alt"
    try:
        env = BatchedDispatchEnv(cost, n2c, off, idx, replicas=R, vehicles=V, depth_limit=cfg["depth"], neighbor_can_server=cfg["neighbor"],
                                 tick_minutes=cfg["tick"], reject_threshold=cfg["threshold"], ring_ticks=cfg["ring_ticks"],
                                 force_generic=fg, idle_cap=idle_cap or max(64, V), ring_cap=max(16, V), far_cap=max(64, V), **kw)
        env.load_orders(rel, pick, dele)
    finally:
        os.environ.pop("VDS_DENSE_DFS", None)
        os.environ.pop("VDS_DENSE_TICK_FORMS", None)
    env.reset(init)
    oracles = []
    for r in range(R):
        o = Oracle(cost, n2c, off, idx, cfg["depth"], cfg["neighbor"], rel, pick, dele, V, tick_minutes=cfg["tick"], reject_threshold=cfg["threshold"])
        o.reset(init[r])
        oracles.append(o)
    assert env.T == oracles[0].num_ticks
    if whole_day:
        env.run(env.T)
        ob = env.obs()
        for r, o in enumerate(oracles):
            o.run_day()
            oo = o.obs()
            for k in ("supply", "idle_now", "inflight"):
                np.testing.assert_array_equal(ob[k][r], oo[k], err_msg="seed %d replica %d %s cfg %s" % (seed, r, k, cfg))
    for t in range(0 if whole_day else env.T):
        env.step()
        for o in oracles:
            o.begin_tick()
        if cfg["dispatch"] and t % 4 == 1:
            reps, cls, poss, tgts = [], [], [], []
            for r, o in enumerate(oracles):
                L = o.lists()
                nidle = int(L["idle_off"][-1])
                if nidle == 0:
                    continue
                pickn = rng.choice(nidle, size=min(nidle, int(rng.integers(1, 5))), replace=False)
                vehs = L["idle_veh"][pickn]
                tg = rng.choice(valid_nodes, size=vehs.size).astype(np.int32)
                for v, flat, tnode in zip(vehs, pickn, tg):
                    c = int(np.searchsorted(L["idle_off"], flat, side="right") - 1)
                    reps.append(r); cls.append(c); poss.append(int(flat - L["idle_off"][c])); tgts.append(int(tnode))
                o.dispatch(vehs, tg)
            if reps and seed % 2 == 1:     # odd seeds: the same actions as a device-resident [R, K, 3] tensor
                import torch
                cntr = np.bincount(reps, minlength=R)
                acts = np.full((R, int(cntr.max()) + 1, 3), -1, dtype=np.int32)
                slot = np.zeros(R, dtype=np.int64)
                for r_, c_, p_, t_ in zip(reps, cls, poss, tgts):
                    acts[r_, slot[r_] + (slot[r_] > 0)] = (c_, p_, t_)      # leave an empty slot after the first action
                    slot[r_] += 1
                held = torch.from_numpy(acts).cuda()
                env.apply_dispatch_torch(held)
                env.sync()
            elif reps:
                env.apply_dispatch(reps, cls, poss, tgts)
        if t % 9 == 0:
            ob = env.obs()
            for r, o in enumerate(oracles):
                oo = o.obs()
                np.testing.assert_array_equal(ob["supply"][r], oo["supply"])
                np.testing.assert_array_equal(ob["idle_now"][r], oo["idle_now"])
                np.testing.assert_array_equal(ob["inflight"][r], oo["inflight"])
        env.advance()
        for o in oracles:
            o.end_tick()
    got, cn = env.orders(), env.counters()
    for r, o in enumerate(oracles):
        exp, oc = o.orders(), o.counters()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][r], exp[k], err_msg="seed %d replica %d %s cfg %s" % (seed, r, k, cfg))
        for i, k in enumerate(("order_num", "reject_num", "matched", "wait_sum", "dispatch_num", "dispatch_cost", "sum_order_value", "evals")):
            assert cn[r, i] == oc[k], (seed, r, k, cfg)
        L, G = o.lists(), env.lists(r)
        for k in ("idle_off", "idle_veh", "arr_off", "arr_veh", "arr_min"):
            np.testing.assert_array_equal(G[k], L[k], err_msg="seed %d replica %d %s" % (seed, r, k))
    env.close()


def volume_worker_argvs():
    import sys
    n_random, n_medium, nproc = int(os.environ.get("VDS_FUZZ_VOLUME", "4000")), int(os.environ.get("VDS_FUZZ_VOLUME_MEDIUM", "128")), int(os.environ.get("VDS_FUZZ_PROCS", "4"))
    here = os.path.dirname(os.path.abspath(__file__))
    return [[sys.executable, os.path.join(here, "fuzz_worker.py"), str(i), str(nproc), str(n_random), str(n_medium)] for i in range(nproc)]


def test_fuzz_volume_in_worker_processes():
    """Volume: VDS_FUZZ_VOLUME (default 4000) more random cities and 128 medium ones, whole days without hooks (run_case(whole_day=True):
    final per-order results, counters, observations and containers of every replica against the oracle), dealt to four worker
    processes (VDS_FUZZ_PROCS; tests/fuzz_worker.py) - the per-case cost is host work (city, oracle, comparisons), which the processes share.  The
    workers are started when the session's collection is final (tests/conftest.py) and run beside the tests before this one."""
    import conftest
    n_random, n_medium = int(os.environ.get("VDS_FUZZ_VOLUME", "4000")), int(os.environ.get("VDS_FUZZ_VOLUME_MEDIUM", "128"))
    conftest.start_workers("fuzz_volume", volume_worker_argvs())
    done = 0
    for i, (rc, out) in enumerate(conftest.collect_workers("fuzz_volume")):
        assert rc == 0 and "WORKER DONE" in out, "worker %d:\n%s" % (i, out[-3000:])
        done += int(out.split("WORKER DONE")[1].split()[0])
    assert done == n_random + n_medium


def days_case(seed):
    """random_case's city with 2-4 order days of different length / density (so different tick grids) and a random
    replica -> day map; dispatches carry explicit arrival minutes and counted flags (vds_apply_dispatch_ex)."""
    cost, n2c, nbr, V, rel, pick, dele, valid_nodes, cfg = random_case(7000 + seed)
    rng = np.random.default_rng(9000 + seed)
    days = [(rel, pick, dele)]
    for d in range(int(rng.integers(1, 4))):
        O = int(rng.integers(2, 1200))
        span = int(rng.integers(20, 1440))
        r2 = (np.sort(rng.integers(0, span, size=O)) + int(rng.integers(0, 50))).astype(np.int32)
        days.append((r2, rng.choice(valid_nodes, size=O).astype(np.int32), rng.choice(valid_nodes, size=O).astype(np.int32)))
    cfg = dict(cfg, R=int(rng.integers(2, 9)))
    rd = rng.integers(0, len(days), size=cfg["R"]).astype(np.int32)
    return cost, n2c, nbr, V, days, rd, valid_nodes, cfg


# 1906: one cluster, 134 vehicles, per-row days - a row without an order at a match step while its 128-slot table is full (the
# round-3 tag-register match loop retired slot 127 of such a row)
DAYS_SEEDS = sorted(set(range(int(os.environ.get("VDS_FUZZ_DAYS_N", "90")))) | {1906})


@pytest.mark.parametrize("seed", DAYS_SEEDS)
def test_random_city_with_replica_days_matches_oracle(seed):
    cost, n2c, nbr, V, days, rd, valid_nodes, cfg = days_case(seed)
    off, idx = neighbors_to_csr(nbr)
    R = cfg["R"]
    rng = np.random.default_rng(2000 + seed)
    init = rng.choice(valid_nodes, size=(R, V)).astype(np.int32) if V else np.zeros((R, 0), np.int32)
    env = BatchedDispatchEnv(cost, n2c, off, idx, replicas=R, vehicles=V, depth_limit=cfg["depth"], neighbor_can_server=cfg["neighbor"],
                             tick_minutes=cfg["tick"], reject_threshold=cfg["threshold"], ring_ticks=cfg["ring_ticks"],
                             force_generic=cfg["force_generic"], idle_cap=max(64, V), ring_cap=max(16, V), far_cap=max(64, V))
    env.load_order_days(days, rd)
    env.reset(init)
    oracles = []
    for r in range(R):
        d = days[rd[r]]
        o = Oracle(cost, n2c, off, idx, cfg["depth"], cfg["neighbor"], d[0], d[1], d[2], V, tick_minutes=cfg["tick"], reject_threshold=cfg["threshold"])
        o.reset(init[r])
        oracles.append(o)
    Ts = [o.num_ticks for o in oracles]
    assert env.T == max(Ts)
    for t in range(env.T):
        env.step()
        live = [r for r in range(R) if t < Ts[r]]
        for r in live:
            oracles[r].begin_tick()
        if cfg["dispatch"] and t % 3 == 1:
            reps, cls, poss, tgts, arrs, cnts = [], [], [], [], [], []
            for r in live:
                o = oracles[r]
                L = o.lists()
                nidle = int(L["idle_off"][-1])
                if nidle == 0:
                    continue
                pickn = rng.choice(nidle, size=min(nidle, int(rng.integers(1, 4))), replace=False)
                vehs = L["idle_veh"][pickn]
                tg = rng.choice(valid_nodes, size=vehs.size).astype(np.int32)
                delay = rng.integers(0, 25, size=vehs.size)
                loc = o.vehicles()["loc"][vehs]
                arr = (o.now_min + cost[tg, loc] + delay).astype(np.int32)          # RoadCost(loc, target) = cost[target, loc]
                counted = bool(rng.random() < 0.5)
                for v, flat, tnode, am in zip(vehs, pickn, tg, arr):
                    c = int(np.searchsorted(L["idle_off"], flat, side="right") - 1)
                    reps.append(r); cls.append(c); poss.append(int(flat - L["idle_off"][c])); tgts.append(int(tnode)); arrs.append(int(am)); cnts.append(int(counted))
                o.dispatch_at(vehs, tg, arrive_min=arr, counted=counted)
            if reps:
                env.apply_dispatch(reps, cls, poss, tgts, arrive_min=arrs, counted=cnts)
        if t % 7 == 0:
            ob = env.obs()
            for r in live:
                oo = oracles[r].obs()
                np.testing.assert_array_equal(ob["supply"][r], oo["supply"])
                np.testing.assert_array_equal(ob["idle_now"][r], oo["idle_now"])
        env.advance()
        for r in live:
            oracles[r].end_tick()
    got, cn = env.orders(), env.counters()
    for r, o in enumerate(oracles):
        exp, oc = o.orders(), o.counters()
        n = exp["status"].size
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][r][:n], exp[k], err_msg="seed %d replica %d %s cfg %s" % (seed, r, k, cfg))
        for i, k in enumerate(("order_num", "reject_num", "matched", "wait_sum", "dispatch_num", "dispatch_cost", "sum_order_value", "evals")):
            assert cn[r, i] == oc[k], (seed, r, k, cfg)
        L, G = o.lists(), env.lists(r)
        for k in ("idle_off", "idle_veh", "arr_off", "arr_veh", "arr_min"):
            np.testing.assert_array_equal(G[k], L[k], err_msg="seed %d replica %d %s" % (seed, r, k))
    env.close()
