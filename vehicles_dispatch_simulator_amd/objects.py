"""Entity views with the field names of the reference's ``objects/objects.py``.

In the reference these are mutable Python objects that *are* the state.  Here the state lives
in HBM as Struct-of-Arrays tables (csrc/vds_device.h); the classes below are thin read views
over a per-tick host snapshot of ONE replica, built lazily by ``Simulation`` so that code written
against the reference's attributes (``cluster.IdleVehicles[i].LocationNode``,
``order.ArriveInfo``, ``len(cluster.VehiclesArrivetime)`` ...) keeps working.  Mutations go
through ``Simulation.DispatchVehicle``.
"""
from __future__ import annotations


class Order(object):
    """Fields of ``objects.py:44-54``."""

    __slots__ = ("_sim", "ID", "ReleasTime", "PickupPoint", "DeliveryPoint", "PickupTimeWindow", "OrderValue")

    def __init__(self, sim, ID, ReleasTime, PickupPoint, DeliveryPoint, PickupTimeWindow, OrderValue):
        self._sim = sim
        self.ID = ID
        self.ReleasTime = ReleasTime
        self.PickupPoint = PickupPoint
        self.DeliveryPoint = DeliveryPoint
        self.PickupTimeWindow = PickupTimeWindow
        self.OrderValue = OrderValue

    @property
    def PickupWaitTime(self):
        st, _, wait = self._sim._order_result(self.ID)
        return int(wait) if st == 1 else None

    @property
    def ArriveInfo(self):
        """None / "Reject" / "Success" / "ArriveTime:<timestamp>" (``objects.py:56-57``,
        ``simulator.py:944,965,1018-1019``)."""
        st, _, wait = self._sim._order_result(self.ID)
        if st == 0:
            return None
        if st == 2:
            return "Reject"
        arrived = self._sim._order_arrival_time(self.ID, int(wait))
        return "Success" if arrived is None else "ArriveTime:" + str(arrived)

    @property
    def Vehicle(self):
        st, veh, _ = self._sim._order_result(self.ID)
        return self._sim.Vehicles[int(veh)] if st == 1 else None

    def Example(self):
        print("Order Example output")
        for k in ("ID", "ReleasTime", "PickupPoint", "DeliveryPoint", "PickupTimeWindow", "PickupWaitTime", "ArriveInfo"):
            print(k + ":", getattr(self, k))
        print()


class Vehicle(object):
    """Fields of ``objects.py:75-82``."""

    __slots__ = ("_sim", "_index", "ID")

    def __init__(self, sim, index, ID):
        self._sim = sim
        self._index = index
        self.ID = ID

    @property
    def LocationNode(self):
        return int(self._sim._mirror()["loc"][self._index])

    @property
    def DeliveryPoint(self):
        d = int(self._sim._mirror()["dest"][self._index])
        return None if d < 0 else d

    @property
    def Cluster(self):
        return self._sim.Clusters[int(self._sim._mirror()["cluster"][self._index])]

    @property
    def Orders(self):
        o = int(self._sim._mirror()["order"][self._index])
        return [] if o < 0 else [self._sim.Orders[o]]

    def Example(self):
        print("Vehicle Example output")
        for k in ("ID", "LocationNode", "Cluster", "Orders", "DeliveryPoint"):
            print(k + ":", getattr(self, k))
        print()


class Cluster(object):
    """Fields of ``objects.py:6-19`` (``Grid`` has the same ones, ``objects.py:131-143``)."""

    def __init__(self, sim, ID, Nodes):
        self._sim = sim
        self.ID = ID
        self.Nodes = Nodes                     # [(node, (lon, lat))]
        self.Neighbor = []                     # filled by Simulation with Cluster objects, reference order
        self.RebalanceNumber = 0
        self.PerRebalanceIdleVehicles = 0
        self.LaterRebalanceIdleVehicles = 0
        self.RebalanceFrequency = 0
        self.PerDispatchIdleVehicles = 0
        self.LaterDispatchIdleVehicles = 0
        self.DispatchNumber = 0

    @property
    def IdleVehicles(self):
        """Vehicles in the reference's list order (``objects.py:10``)."""
        L = self._sim._lists()
        a, b = L["idle_off"][self.ID], L["idle_off"][self.ID + 1]
        V = self._sim.Vehicles
        return [V[int(v)] for v in L["idle_veh"][a:b]]

    @property
    def VehiclesArrivetime(self):
        """{Vehicle: arrival Timestamp} in dict insertion order (``objects.py:11``)."""
        L = self._sim._lists()
        a, b = L["arr_off"][self.ID], L["arr_off"][self.ID + 1]
        V = self._sim.Vehicles
        return {V[int(v)]: self._sim._minute_to_time(int(m)) for v, m in zip(L["arr_veh"][a:b], L["arr_min"][a:b])}

    @property
    def Orders(self):
        """Orders released in the current slot whose pickup is in this cluster (``simulator.py:919``)."""
        return [self._sim.Orders[i] for i in self._sim._tick_orders_of_cluster(self.ID)]

    @property
    def PerMatchIdleVehicles(self):
        return int(self._sim._obs()["idle_pre"][self.ID])

    def Example(self):
        print("ID:", self.ID)
        print("Nodes:", self.Nodes)
        print("Neighbor:[", " ".join(str(i.ID) for i in self.Neighbor), "]")
        print("RebalanceNumber:", self.RebalanceNumber)
        print("IdleVehicles:", self.IdleVehicles)
        print("VehiclesArrivetime:", self.VehiclesArrivetime)
        print("Orders:", self.Orders)
        print()


class Grid(Cluster):
    pass


class Transition(object):
    """Plain record for RL agents (``objects.py:105-128``); never touched by the tick path."""

    def __init__(self, FromCluster, ArriveCluster, Vehicle, State, StateQTable, Action, TotallyReward, PositiveReward,
                 NegativeReward, NeighborNegativeReward, State_, State_QTable):
        self.FromCluster = FromCluster
        self.ArriveCluster = ArriveCluster
        self.Vehicle = Vehicle
        self.State = State
        self.StateQTable = StateQTable
        self.Action = Action
        self.TotallyReward = TotallyReward
        self.PositiveReward = PositiveReward
        self.NegativeReward = NegativeReward
        self.NeighborNegativeReward = NeighborNegativeReward
        self.State_ = State_
        self.State_QTable = State_QTable

    def Example(self):
        print("Transition Example output")
        for k in ("Action", "TotallyReward", "PositiveReward", "NegativeReward", "NeighborNegativeReward"):
            print(k + ":", getattr(self, k))
        print()
