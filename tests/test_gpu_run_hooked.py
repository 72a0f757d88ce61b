"""vds_run_hooked: SimCity's loop with the dispatch hook on the device as ONE graph launch per day (reference simulator.py:1048-1091:
step -> observations -> policy -> DispatchFunction body -> advance), against the same loop issued call by call - whose every piece
the other parity files check against the oracle - and against an oracle replay of the actions.  Plain tick and neighbour search, one
group and several replica groups (parallel branches), with a torch policy embedded as a child graph and with a fixed action tensor."""
import numpy as np
import pytest

from helpers import engine_settings, load_golden, make_oracle
from vehicles_dispatch_simulator_amd import BatchedDispatchEnv

pytestmark = pytest.mark.gpu


def make_env(g, R, stream):
    env = BatchedDispatchEnv(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], replicas=R, vehicles=int(g["V"]),
                             depth_limit=int(g["depth_limit"]), neighbor_can_server=bool(g["neighbor_can_server"]), stream=stream, **engine_settings(g))
    env.load_orders(g["o_release_min"], g["o_pickup"], g["o_delivery"])
    return env


def state_of(env):
    env.sync()
    od, cn = env.orders(), env.counters()
    lists = [env.lists(r) for r in (0, env.R // 2, env.R - 1)]
    return od, cn, lists


def assert_same(a, b):
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(a[0][k], b[0][k], err_msg=k)
    np.testing.assert_array_equal(a[1], b[1])
    for la, lb in zip(a[2], b[2]):
        for k in ("idle_off", "idle_veh", "arr_off", "arr_veh", "arr_min"):
            np.testing.assert_array_equal(la[k], lb[k], err_msg=k)


@pytest.mark.parametrize("name,groups", [("tiny_kmeans", 1), ("tiny_kmeans", 3), ("tiny_kmeans_dfs2", 1), ("tiny_kmeans_dfs2", 2), ("tiny_grid", 2)])
def test_hooked_day_graph_with_a_torch_policy_equals_the_stepwise_loop_and_the_oracle(name, groups):
    import torch
    g = load_golden(name)
    R, K = 48, 3
    stream = torch.cuda.current_stream()
    n2c = np.asarray(g["node2cluster"])
    C = int(g["C"])
    some_node_of = torch.tensor([int(np.flatnonzero(n2c == c)[0]) if (n2c == c).any() else int(np.flatnonzero(n2c >= 0)[0]) for c in range(C)], dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(5)
    init = np.stack([g["veh_node"]] + [rng.permutation(g["veh_node"]) for _ in range(R - 1)]).astype(np.int32)

    def policy(ob):     # move one idle vehicle from each of the K fullest clusters to the K emptiest (skipped where the list is empty)
        idle = ob[1]
        src = torch.topk(idle, K, dim=1).indices
        dst = torch.topk(idle + ob[2], K, dim=1, largest=False).indices
        ok = idle.gather(1, src) > 1
        return torch.stack([torch.where(ok, src.int(), torch.full_like(src, -1).int()), torch.zeros_like(src).int(), some_node_of[dst]], dim=2).contiguous()

    # (a) call by call, eager policy; the actions are logged for the oracle
    env_a = make_env(g, R, stream.cuda_stream)
    env_a.reset(init)
    log = []
    for t in range(env_a.T):
        env_a.step()
        acts = policy(env_a.obs_torch(inflight=False))
        log.append((acts.cpu().numpy(), [env_a.lists(r) for r in (0, R - 1)]))
        env_a.apply_dispatch_torch(acts)
        env_a.advance()
    ref = state_of(env_a)
    # (b) one graph launch per day, the policy captured once and embedded as a child graph
    env_b = make_env(g, R, stream.cuda_stream)
    env_b.set_run_groups(groups, -1)
    env_b.reset(init)
    obs_static = env_b.obs_torch(inflight=False)
    actions = torch.zeros((R, K, 3), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        for _ in range(2):
            actions.copy_(policy(obs_static))
    stream.wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph):
        actions.copy_(policy(obs_static))
    for day in range(2):        # (the second day replays the executable graph)
        env_b.reset_again()
        env_b.run_hooked(env_b.T, actions=actions, policy_graph=graph)
        assert_same(state_of(env_b), ref)
    # in two halves: another first slot, another length - rebuilt / updated in place
    env_b.reset_again()
    env_b.run_hooked(50, actions=actions, policy_graph=graph)
    env_b.run_hooked(env_b.T - 50, actions=actions, policy_graph=graph)
    assert_same(state_of(env_b), ref)
    # the oracle replaying the logged actions of two replicas
    for ri, r in enumerate((0, R - 1)):
        o = make_oracle(g)
        o.reset(init[r])
        for t in range(env_a.T):
            o.begin_tick()
            acts, lists = log[t]
            L = lists[ri]
            vehs, tgts = [], []
            for k in range(K):
                cl, pos, tgt = acts[r, k]
                if cl >= 0:
                    vehs.append(L["idle_veh"][L["idle_off"][cl] + pos]); tgts.append(tgt)
            if vehs:
                o.dispatch(np.array(vehs), np.array(tgts))
            o.end_tick()
        oo = o.orders()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(ref[0][k][r], oo[k], err_msg="replica %d %s" % (r, k))
    env_a.close(); env_b.close()


def test_hooked_day_with_fixed_actions_no_policy_and_observation_subsets():
    """No policy graph: the tensor is applied as it stands every slot; observation planes on demand; without actions the run equals vds_run."""
    import torch
    g = load_golden("tiny_kmeans")
    R = 40
    stream = torch.cuda.current_stream()
    init = np.tile(g["veh_node"], (R, 1)).astype(np.int32)
    n2c = np.asarray(g["node2cluster"])
    acts = np.full((R, 2, 3), -1, dtype=np.int32)
    acts[:, 0] = (np.arange(R) % int(g["C"]))[:, None] * np.array([1, 0, 0]) + np.array([0, 0, int(np.flatnonzero(n2c == 3)[0])])
    actions = torch.from_numpy(acts).cuda()
    env_a = make_env(g, R, stream.cuda_stream); env_a.reset(init)
    for t in range(env_a.T):
        env_a.step(); env_a.apply_dispatch_torch(actions); env_a.advance()
    try:
        env_a.sync()
    except Exception as e:      # (moves from empty lists are skipped and reported once)
        assert "skipped" in str(e)
    ref = state_of(env_a)
    env_b = make_env(g, R, stream.cuda_stream); env_b.set_run_groups(2, -1); env_b.reset(init)
    env_b.run_hooked(env_b.T, actions=actions, idle_pre=False, supply=False, cl_orders=False)
    try:
        env_b.sync()
    except Exception as e:
        assert "skipped" in str(e)
    assert_same(state_of(env_b), ref)
    # the observation block holds the last slot's idle_now plane
    np.testing.assert_array_equal(env_b.obs_torch(idle_pre=False, supply=False, cl_orders=False, inflight=False)[1].cpu().numpy(), env_a.obs()["idle_now"])
    # no actions, no planes: vds_run
    env_c = make_env(g, R, stream.cuda_stream); env_c.reset(init); env_c.run(env_c.T)
    env_b.reset_again(); env_b.run_hooked(env_b.T, actions=None, idle_pre=False, idle_now=False, supply=False, cl_orders=False)
    assert_same(state_of(env_b), state_of(env_c))
    with pytest.raises(Exception, match="past the end"):
        env_b.run_hooked(1)
    for e in (env_a, env_b, env_c):
        e.close()


@pytest.mark.parametrize("name", ["tiny_kmeans", "tiny_kmeans_dfs2"])
def test_observations_in_place_are_the_packed_planes_and_feed_a_hooked_day_without_an_observation_pass(name):
    """vds_obs_inplace: idle_pre / idle_now / cl_orders read where the tick keeps them (strided [R, C] views of the bucket records)
    equal the packed planes after every step and after every dispatch; a policy graph over the views with NO observation planes in
    the hooked day (planes = 0: no k_pack_obs per slot) gives the day of the stepwise loop over the packed block."""
    import torch
    g = load_golden(name)
    R, K = 24, 2
    stream = torch.cuda.current_stream()
    n2c = np.asarray(g["node2cluster"])
    C = int(g["C"])
    some_node_of = torch.tensor([int(np.flatnonzero(n2c == c)[0]) if (n2c == c).any() else int(np.flatnonzero(n2c >= 0)[0]) for c in range(C)], dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(9)
    init = np.stack([rng.permutation(g["veh_node"]) for _ in range(R)]).astype(np.int32)

    def policy(idle, orders):
        src = torch.topk(idle - orders, K, dim=1).indices
        dst = torch.topk(idle - orders, K, dim=1, largest=False).indices
        ok = idle.gather(1, src) > 1
        return torch.stack([torch.where(ok, src.int(), torch.full_like(src, -1).int()), torch.zeros_like(src).int(), some_node_of[dst]], dim=2).contiguous()

    env_a = make_env(g, R, stream.cuda_stream)
    env_a.reset(init)
    views = env_a.obs_inplace_torch()
    assert all(v.shape == (R, C) and v.dtype == torch.int32 and v.is_cuda for v in views.values())
    for t in range(env_a.T):
        env_a.step()
        packed = env_a.obs_torch(inflight=False)
        for i, k in ((0, "idle_pre"), (1, "idle_now"), (3, "cl_orders")):
            assert torch.equal(views[k], packed[i]), (t, k)
        env_a.apply_dispatch_torch(policy(packed[1], packed[3]))
        assert torch.equal(views["idle_now"], env_a.obs_torch(idle_pre=False, supply=False, cl_orders=False, inflight=False)[1]), t
        env_a.advance()
    ref = state_of(env_a)
    env_b = make_env(g, R, stream.cuda_stream)
    env_b.reset(init)
    vb = env_b.obs_inplace_torch()
    actions = torch.zeros((R, K, 3), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        for _ in range(2):
            actions.copy_(policy(vb["idle_now"], vb["cl_orders"]))
    stream.wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph):
        actions.copy_(policy(vb["idle_now"], vb["cl_orders"]))
    for day in range(2):
        env_b.reset_again()
        env_b.run_hooked(env_b.T, actions=actions, policy_graph=graph, idle_pre=False, idle_now=False, supply=False, cl_orders=False)
        assert_same(state_of(env_b), ref)
    env_a.close(); env_b.close()


def test_observations_in_place_are_refused_when_replicas_are_stored_by_order_day():
    from vehicles_dispatch_simulator_amd import workloads
    w = workloads.tiny(vehicles=150, orders=1500)
    days = workloads.distinct_days(w, 3)
    env = w.make_env(48, load=False)
    env.load_order_days(days, (np.arange(48) % 3).astype(np.int32))
    env.reset(w.vehicle_nodes(48))
    with pytest.raises(Exception, match="regrouped"):
        env.obs_inplace_torch()
    with pytest.raises(Exception, match="not kept in place"):
        import ctypes as C
        p, a, b = C.c_void_p(), C.c_int64(), C.c_int64()
        env._chk(env._lib.vds_obs_inplace(env._h, 2, C.byref(p), C.byref(a), C.byref(b)))
    env.close()
