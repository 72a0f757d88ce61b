"""Entity views with the field names, setters and mutators of the reference's ``objects/objects.py``.

In the reference these are mutable Python objects that *are* the state.  Here the state lives
in HBM as Struct-of-Arrays tables (csrc/vds_device.h); the classes below are views over a
per-tick host snapshot of ONE replica, kept by ``Simulation``, so that code written against the
reference's attributes (``cluster.IdleVehicles[i].LocationNode``, ``order.ArriveInfo``,
``len(cluster.VehiclesArrivetime)`` ...) keeps working - including code that WRITES them the way
a ``DispatchFunction`` body does in the reference's idiom:

    veh.DeliveryPoint = node
    veh.Cluster.IdleVehicles.remove(veh)
    sim.NodeID2Cluseter[node].VehiclesArrivetime[veh] = sim.RealExpTime + np.timedelta64(cost * MINUTES)
    sim.DispatchNum += 1; sim.TotallyDispatchCost += cost

``Cluster.IdleVehicles`` is a real ``list`` and ``Cluster.VehiclesArrivetime`` a real ``dict``, one
permanent object per cluster (refreshed in place after every device step, as the reference's
containers are permanent), so the edits above are ordinary Python; when the hook returns,
``Simulation`` diffs them against the snapshot and turns them into ``vds_apply_dispatch_ex``
actions (INTEGRATION.md section 3 lists what can and cannot be expressed on the device).
"""
from __future__ import annotations


_UNSET = object()


class Order(object):
    """Fields and methods of ``objects.py:44-72``.  ``PickupWaitTime`` / ``ArriveInfo`` are computed from the device's
    per-order results unless user code assigned them (host-side override, cleared by ``Simulation.Reset``)."""

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
        ov = self._sim._order_override.get((self.ID, "PickupWaitTime"), _UNSET)
        if ov is not _UNSET:
            return ov
        st, _, wait = self._sim._order_result(self.ID)
        return int(wait) if st == 1 else None

    @PickupWaitTime.setter
    def PickupWaitTime(self, value):
        self._sim._order_override[(self.ID, "PickupWaitTime")] = value

    @property
    def ArriveInfo(self):
        """None / "Reject" / "Success" / "ArriveTime:<timestamp>" (``objects.py:56-57``,
        ``simulator.py:944,965,1018-1019``)."""
        ov = self._sim._order_override.get((self.ID, "ArriveInfo"), _UNSET)
        if ov is not _UNSET:
            return ov
        st, _, wait = self._sim._order_result(self.ID)
        if st == 0:
            return None
        if st == 2:
            return "Reject"
        arrived = self._sim._order_arrival_time(self.ID, int(wait))
        return "Success" if arrived is None else "ArriveTime:" + str(arrived)

    @ArriveInfo.setter
    def ArriveInfo(self, value):
        self._sim._order_override[(self.ID, "ArriveInfo")] = value

    def ArriveOrderTimeRecord(self, ArriveTime):
        """``objects.py:56-57``."""
        self.ArriveInfo = "ArriveTime:" + str(ArriveTime)

    def Reset(self):
        """``objects.py:70-72``."""
        self.PickupWaitTime = None
        self.ArriveInfo = None

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
    """Fields and methods of ``objects.py:75-102``.  Reads come from the host mirror of the replica; writes update the
    mirror and - for ``DeliveryPoint`` - feed the dispatch the hook body is composing (``Simulation._flush_dispatch``)."""

    __slots__ = ("_sim", "_index", "ID")

    def __init__(self, sim, index, ID):
        self._sim = sim
        self._index = index
        self.ID = ID

    @property
    def LocationNode(self):
        return int(self._sim._mirror()["loc"][self._index])

    @LocationNode.setter
    def LocationNode(self, node):
        self._sim._mirror()["loc"][self._index] = int(node)

    @property
    def DeliveryPoint(self):
        d = int(self._sim._mirror()["dest"][self._index])
        return None if d < 0 else d

    @DeliveryPoint.setter
    def DeliveryPoint(self, node):
        self._sim._mirror()["dest"][self._index] = -1 if node is None else int(node)

    @property
    def Cluster(self):
        return self._sim.Clusters[int(self._sim._mirror()["cluster"][self._index])]

    @Cluster.setter
    def Cluster(self, cluster):
        self._sim._mirror()["cluster"][self._index] = int(cluster.ID)

    @property
    def Orders(self):
        o = int(self._sim._mirror()["order"][self._index])
        return [] if o < 0 else [self._sim.Orders[o]]

    def ArriveVehicleUpDate(self, DeliveryCluster):
        """``objects.py:84-89`` on the host view.  (The device performs this itself in UpdateFunction; a hook that
        forces an early arrival cannot be mirrored to the device and is refused when the hook returns.)"""
        self.LocationNode = self.DeliveryPoint
        self.DeliveryPoint = None
        self.Cluster = DeliveryCluster
        self._sim._mirror()["order"][self._index] = -1

    def Reset(self):
        """``objects.py:91-93``."""
        self._sim._mirror()["order"][self._index] = -1
        self.DeliveryPoint = None

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
        self._per_match_override = None

    @property
    def IdleVehicles(self):
        """Vehicles in the reference's list order (``objects.py:10``): a real, permanent ``list`` - edits made inside a
        hook (``remove`` / ``pop`` / ``del``) are applied to the device when the hook returns."""
        return self._sim._idle_container(self.ID)

    @IdleVehicles.setter
    def IdleVehicles(self, value):
        lst = self._sim._idle_container(self.ID)
        lst[:] = list(value)

    @property
    def VehiclesArrivetime(self):
        """{Vehicle: arrival Timestamp} in dict insertion order (``objects.py:11``): a real, permanent ``dict`` -
        ``d[veh] = time`` inside a hook becomes the arrival entry of a dispatch."""
        return self._sim._arrival_container(self.ID)

    @VehiclesArrivetime.setter
    def VehiclesArrivetime(self, value):
        d = self._sim._arrival_container(self.ID)
        items = list(value.items())
        d.clear()
        d.update(items)

    def Reset(self):
        """``objects.py:20-27`` on the host view (``Simulation.Reset`` follows it with a device reset)."""
        self.RebalanceNumber = 0
        self.IdleVehicles.clear()
        self.VehiclesArrivetime.clear()
        self.PerRebalanceIdleVehicles = 0
        self._per_match_override = 0

    def ArriveClusterUpDate(self, vehicle):
        """``objects.py:29-31`` on the host view (see ``Vehicle.ArriveVehicleUpDate``)."""
        self.IdleVehicles.append(vehicle)
        self.VehiclesArrivetime.pop(vehicle)

    @property
    def Orders(self):
        """Orders released in the current slot whose pickup is in this cluster (``simulator.py:919``)."""
        return [self._sim.Orders[i] for i in self._sim._tick_orders_of_cluster(self.ID)]

    @property
    def PerMatchIdleVehicles(self):
        if self._per_match_override is not None:
            return self._per_match_override
        return int(self._sim._obs()["idle_pre"][self.ID])

    @PerMatchIdleVehicles.setter
    def PerMatchIdleVehicles(self, value):
        self._per_match_override = value

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
