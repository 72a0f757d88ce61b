"""the objects are MUTABLE, as in the reference (objects/objects.py).

One ``DispatchFunction`` body, written ONCE in the reference's idiom (oracle/dispatch_idiom.py: ``IdleVehicles.remove``,
``VehiclesArrivetime[veh] = time``, ``veh.DeliveryPoint = node``, ``self.DispatchNum += 1`` ...), was executed by the
unmodified reference inside oracle/ref_harness.py when the golden fixtures were made; here the SAME function object runs
inside the GPU shell and must give the identical day: per-tick idle lists in list order, counters, per-order results,
``DispatchNum`` / ``TotallyDispatchCost`` as the body counted them.  Also: the reference's mutators exist and behave,
unsupported edits are refused loudly, the vehicle views are right after a fast-forwarded day, overriding a
fused phase is refused, duplicate dispatches are refused without corrupting the episode, and an
idle-table overflow is survived by regrowing the tables."""
import random

import numpy as np
import pandas as pd
import pytest

from helpers import load_golden, make_oracle
from oracle.dispatch_idiom import move_idle_vehicle, policy_factory
from test_gpu_simulation_shell import make_sim
from vehicles_dispatch_simulator_amd.config.setting import MINUTES
from vehicles_dispatch_simulator_amd.simulation import Simulation

pytestmark = pytest.mark.gpu


class IdiomSim(Simulation):
    """A user subclass exactly as it would be written against the reference."""
    extra_minutes = 0

    def DispatchFunction(self):
        self.seen_idle.append([[v._index for v in c.IdleVehicles] for c in self.Clusters])
        for veh, target in self.policy(self, self.step):
            move_idle_vehicle(self, veh, target, MINUTES, self.extra_minutes)


@pytest.mark.parametrize("name", ["tiny_dispatch", "tiny_dispatch_dfs2", "tiny_dispatch_delay"])
def test_reference_idiom_hook_body_reproduces_reference_day(name):
    g = load_golden(name)
    sim = make_sim(g, IdiomSim)
    sim.policy = policy_factory(int(g["N"]))
    sim.extra_minutes = int(g["dispatch_extra_minutes"]) if "dispatch_extra_minutes" in g else 0
    sim.seen_idle = []
    sim.SimCity()
    assert sim.step == int(g["n_ticks"])
    assert (sim.OrderNum, sim.RejectNum, sim.TotallyWaitTime) == (int(g["order_num"]), int(g["reject_num"]), int(g["wait_sum"]))
    assert (sim.DispatchNum, sim.TotallyDispatchCost) == (int(g["dispatch_num"]), int(g["dispatch_cost"]))   # counted by the body itself
    assert sim.SumOrderValue == int(g["sum_order_value"])
    for t in range(int(g["n_ticks"])):
        off = g["l_idle_off"][t]
        assert sim.seen_idle[t] == [g["l_idle_veh"][t][off[c]:off[c + 1]].tolist() for c in range(int(g["C"]))], t
    got = sim.env.orders(0, 1)
    np.testing.assert_array_equal(got["status"][0], g["o_status"])
    np.testing.assert_array_equal(got["vehicle"][0], g["o_vehicle"])
    np.testing.assert_array_equal(got["wait"][0], g["o_wait"])
    # the device counted nothing itself: DispatchNum lives in the body, as in the reference
    assert sim.env.counters()[0, 4] == 0


class MixedSim(Simulation):
    """Idiom edits and DispatchVehicle calls in one hook; held container references stay valid across slots."""

    def DispatchFunction(self):
        if self.step == 0:
            self.held = [c.IdleVehicles for c in self.Clusters]          # taken once, used in every later slot
            self.held_arr = [c.VehiclesArrivetime for c in self.Clusters]
        for c, lst, d in zip(self.Clusters, self.held, self.held_arr):
            assert lst is c.IdleVehicles and d is c.VehiclesArrivetime
        moves = self.policy(self, self.step)
        for i, (veh, target) in enumerate(moves):
            if i % 2:
                self.DispatchVehicle(veh, target)
            else:
                move_idle_vehicle(self, veh, target, MINUTES)


def test_mixed_idiom_and_api_dispatch_with_held_references():
    g = load_golden("tiny_dispatch")
    sim = make_sim(g, MixedSim)
    sim.policy = policy_factory(int(g["N"]))
    sim.SimCity()
    # same moves, same times: the day is the golden day whichever way each move was expressed
    assert (sim.OrderNum, sim.RejectNum, sim.TotallyWaitTime) == (int(g["order_num"]), int(g["reject_num"]), int(g["wait_sum"]))
    assert (sim.DispatchNum, sim.TotallyDispatchCost) == (int(g["dispatch_num"]), int(g["dispatch_cost"]))
    got = sim.env.orders(0, 1)
    np.testing.assert_array_equal(got["vehicle"][0], g["o_vehicle"])
    assert 0 < sim.env.counters()[0, 4] < int(g["dispatch_num"])


def test_reference_mutators_and_refused_edits():
    g = load_golden("tiny_kmeans")

    class Probe(Simulation):
        def DispatchFunction(self):
            if self.step != 5:
                return
            c = next(c for c in self.Clusters if len(c.IdleVehicles) >= 2 and len(c.VehiclesArrivetime) >= 1)
            self.probe_cluster = c
            if self.mode == "reorder":
                c.IdleVehicles.reverse()
            elif self.mode == "vanish":
                c.IdleVehicles.pop(0)
            elif self.mode == "early_arrival":
                veh = next(iter(c.VehiclesArrivetime))
                c.ArriveClusterUpDate(veh)                      # objects.py:29-31
                veh.ArriveVehicleUpDate(c)                      # objects.py:84-89
            elif self.mode == "wrong_cluster":
                veh = c.IdleVehicles[0]
                other = next(k for k in self.Clusters if k is not c and len(k.Nodes))
                veh.DeliveryPoint = c.Nodes[0][0]
                c.IdleVehicles.remove(veh)
                other.VehiclesArrivetime[veh] = self.RealExpTime
            elif self.mode == "twice":
                veh = c.IdleVehicles[0]
                self.DispatchVehicle(veh, c.Nodes[0][0]); self.DispatchVehicle(veh, c.Nodes[0][0])

    for mode, msg in (("reorder", "reordered or extended"), ("vanish", "entered into no"), ("early_arrival", "reordered or extended|deleted from"),
                      ("wrong_cluster", "not a node of that cluster"), ("twice", "dispatched twice")):
        sim = make_sim(g, Probe)
        sim.mode = mode
        with pytest.raises(Exception, match=msg):
            sim.SimCity()
    # host-side mutators with the reference's semantics
    sim = make_sim(g, Simulation)
    sim.SimCity()
    o = sim.Orders[10]
    assert o.ArriveInfo is not None
    o.ArriveOrderTimeRecord(pd.Timestamp("2016-11-01 10:00"))           # objects.py:56-57
    assert o.ArriveInfo == "ArriveTime:2016-11-01 10:00:00"
    o.Reset()                                                            # objects.py:70-72
    assert o.ArriveInfo is None and o.PickupWaitTime is None
    v = sim.Vehicles[3]
    v.DeliveryPoint = 7; v.LocationNode = 9; v.Cluster = sim.Clusters[2]
    assert (v.DeliveryPoint, v.LocationNode, v.Cluster.ID) == (7, 9, 2)
    v.Reset()                                                            # objects.py:91-93
    assert v.DeliveryPoint is None and v.Orders == []
    c = sim.Clusters[1]
    c.RebalanceNumber = 4
    c.Reset()                                                            # objects.py:20-27
    assert c.RebalanceNumber == 0 and c.IdleVehicles == [] and c.VehiclesArrivetime == {} and c.PerMatchIdleVehicles == 0
    sim.Reset()                                                          # simulator.py:214-247: device and views start over
    assert sim.Orders[10].ArriveInfo is None and sum(len(c.IdleVehicles) for c in sim.Clusters) == len(sim.Vehicles)


def test_vehicle_views_after_fast_forward_match_hooked_loop():
    """SimCity() without hooks runs the day as one device call; the vehicle views must still be those of the
    per-slot loop (LocationNode / Cluster of a vehicle on the way stay at the trip origin, objects.py:84-89)."""
    for name in ("tiny_kmeans", "tiny_kmeans_dfs2"):
        g = load_golden(name)
        a, b = make_sim(g, Simulation), make_sim(g, Simulation)
        a.SimCity()                    # fast-forward
        b.SimCity(FastForward=False)
        o = make_oracle(g); o.run_day()
        veh = o.vehicles()
        for sim in (a, b):
            np.testing.assert_array_equal(np.array([v.LocationNode for v in sim.Vehicles]), veh["loc"])
            np.testing.assert_array_equal(np.array([v.Cluster.ID for v in sim.Vehicles]), veh["cluster"])
            np.testing.assert_array_equal(np.array([-1 if v.DeliveryPoint is None else v.DeliveryPoint for v in sim.Vehicles]), veh["dest"])
            np.testing.assert_array_equal(np.array([v.Orders[0].ID if v.Orders else -1 for v in sim.Vehicles]), veh["order"])
        for ca, cb in zip(a.Clusters, b.Clusters):
            assert [v._index for v in ca.IdleVehicles] == [v._index for v in cb.IdleVehicles]
            assert ca.IdleVehicles and ca.IdleVehicles[0].LocationNode == cb.IdleVehicles[0].LocationNode or not cb.IdleVehicles
        assert (a.OrderNum, a.RejectNum) == (b.OrderNum, b.RejectNum)


def test_overriding_a_fused_phase_is_refused():
    g = load_golden("tiny_kmeans")

    class Bad(Simulation):
        def MatchFunction(self):
            return

    sim = make_sim(g, Bad)
    with pytest.raises(Exception, match="MatchFunction overridden"):
        sim.SimCity()
    plain = make_sim(g, Simulation)
    assert plain.Clusters[0].Orders == []            # readable before SimCity (_stepped_current initialised)


def test_duplicate_dispatch_is_refused_and_episode_survives():
    """the same idle position twice must not put a vehicle into two arrival tables, and the error must not
    poison the rest of the episode."""
    import torch
    g = load_golden("tiny_kmeans")
    from test_gpu_parity import make_env
    R, V = 3, int(g["V"])
    env = make_env(g, R, stream=torch.cuda.current_stream().cuda_stream)
    init = np.tile(g["veh_node"], (R, 1)).astype(np.int32)
    env.reset(init)
    for _ in range(4):
        env.step(); env.advance()
    env.step()
    ob = env.obs()
    c = int(np.argmax(ob["idle_now"][0]))
    node = int(np.flatnonzero(g["node2cluster"] >= 0)[0])
    with pytest.raises(Exception, match="same idle position"):
        env.apply_dispatch([0, 0], [c, c], [1, 1], [node, node])
    env.sync()                                         # nothing was launched: no device error either
    acts = np.full((R, 4, 3), -1, dtype=np.int32)
    acts[1, 0] = (c, 0, node); acts[1, 2] = (c, 0, node)          # replica 1 names position 0 twice
    acts[2, 1] = (c, 0, node)
    env.apply_dispatch_torch(torch.from_numpy(acts).cuda())
    with pytest.raises(Exception, match="listed twice"):
        env.sync()
    env.sync()                                         # reported once, then cleared
    ob2 = env.obs()
    assert ob2["idle_now"][1, c] == ob["idle_now"][1, c] - 1 and ob2["idle_now"][2, c] == ob["idle_now"][2, c] - 1
    assert (ob2["idle_now"].sum(axis=1) + ob2["inflight"].sum(axis=1) == V).all()      # no vehicle is in two containers
    assert env.counters()[1, 4] == 1
    env.advance()
    for _ in range(env.T - 5):
        env.step(); env.advance()
    env.sync()
    fin = env.obs()
    assert (fin["idle_now"].sum(axis=1) + fin["inflight"].sum(axis=1) == V).all()
    env.close()


def test_idle_table_overflow_is_survived_by_regrowing():
    """a day in which vehicles concentrate beyond the table capacity.  Explicit small cap -> reported;
    vds_set_idle_cap + vds_reset_again -> the same day completes, bit-exact against the oracle; lists longer than the
    1024-entry register tables work (no hard limit any more)."""
    from vehicles_dispatch_simulator_amd import BatchedDispatchEnv, workloads
    w = workloads.tiny(vehicles=1500, orders=2500)
    R = 2
    init = w.vehicle_nodes(R)
    node = int(np.flatnonzero(w.city.node2cluster == 3)[0])
    init[1, :1300] = node                              # 1300 vehicles start on one node: one list of > 1024 entries
    env = w.make_env(R, idle_cap=192)
    with pytest.raises(Exception, match="idle table overflow"):
        env.reset(init)
    env.set_idle_cap(1400)
    assert env.idle_cap >= 1400
    env.reset(init)
    env.run(env.T); env.sync()
    from oracle.oracle import Oracle
    got = env.orders()
    for r in range(R):
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server,
                   w.release_min, w.pickup, w.delivery, w.vehicles)
        o.reset(init[r]); o.run_day()
        exp = o.orders()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][r], exp[k], err_msg="replica %d %s" % (r, k))
    env.close()
    # automatic capacity: sized from the start histogram at vds_reset, no error at all
    env2 = w.make_env(R)
    env2.reset(init)
    assert env2.idle_cap >= 1300
    env2.run(env2.T); env2.sync()
    np.testing.assert_array_equal(env2.orders()["vehicle"], got["vehicle"])
    env2.close()
