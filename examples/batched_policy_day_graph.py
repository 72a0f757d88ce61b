"""An agent written against the reference's ``Simulation`` class whose dispatch policy runs INSIDE the engine's day graph:
``BatchedHooks=True`` + ``BatchedPolicy``.  R cities advance together; per slot the engine's tick, the
observation planes, the policy (pure torch, captured once) and the dispatch run on the GPU, and ``SimCity()`` is one graph launch
per day - the host comes back when the day is over.

    python examples/batched_policy_day_graph.py [replicas]

Uses the small synthetic city of ``examples/demo_simulation.py`` (written to a temporary directory in the reference's ./data layout)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch

from vehicles_dispatch_simulator_amd.config.setting import *          # noqa: F401,F403  (same names as the reference)
from vehicles_dispatch_simulator_amd.simulation import Simulation
from demo_simulation import write_synthetic_data_dir


class RebalancingAgent(Simulation):
    """Every slot each city sends the head of the idle lists of its K richest clusters (idle + expected arrivals - waiting orders)
    one vehicle each towards its K poorest."""
    K = 4
    BatchedPolicyPlanes = ("idle_now", "supply", "cl_orders")

    def BatchedPolicy(self, ob):
        surplus = ob["idle_now"] + ob["supply"] - ob["cl_orders"]                     # [R, C] int32
        src = torch.topk(surplus, self.K, dim=1).indices
        dst = torch.topk(surplus, self.K, dim=1, largest=False).indices
        ok = (ob["idle_now"].gather(1, src) > 0) & (surplus.gather(1, src) - surplus.gather(1, dst) > 2)
        acts = torch.stack([torch.where(ok, src, torch.full_like(src, -1)).int(), torch.zeros_like(src).int(), self.node_of_cluster[dst]], dim=2)
        self.moves += ok.sum()                                                        # (device state, updated in place: capturable)
        return acts.contiguous()

    def BatchedPolicyBegin(self):
        self.moves.zero_()


if __name__ == "__main__":
    os.environ.setdefault("TZ", "UTC")
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    data_dir = write_synthetic_data_dir(tempfile.mkdtemp(prefix="vds_policy_"))
    sim = RebalancingAgent(ClusterMode=ClusterMode, DemandPredictionMode=DemandPredictionMode, DispatchMode=DispatchMode, VehiclesNumber=300,
                           TimePeriods=TIMESTEP, LocalRegionBound=LocalRegionBound, SideLengthMeter=2400, VehiclesServiceMeter=VehiclesServiceMeter,
                           NeighborCanServer=NeighborCanServer, FocusOnLocalRegion=FocusOnLocalRegion, DataDir=data_dir, Replicas=R, VehicleSeed=7,
                           BatchedHooks=True, Quiet=True)
    sim.CreateAllInstantiate()
    n2c = np.asarray(sim._world.node2cluster)
    C = sim.env.C
    sim.node_of_cluster = torch.tensor([int(np.flatnonzero(n2c == c)[0]) if (n2c == c).any() else 0 for c in range(C)], dtype=torch.int32, device="cuda")
    sim.moves = torch.zeros((), dtype=torch.int64, device="cuda")
    for episode in range(3):
        if episode:
            sim.Reset()
        t0 = time.perf_counter()
        sim.SimCity()
        dt = time.perf_counter() - t0
        cn = sim.BatchedCounters().cpu().numpy()
        print("episode %d: %d cities x %d slots in %.1f ms (%s); dispatches %d, rejects per city %.1f, policy said 'move' %d times" % (
            episode, R, sim.env.T, dt * 1e3, "one graph launch" if sim.BatchedPolicyGraphError is None else "slot by slot: " + sim.BatchedPolicyGraphError,
            int(cn[:, 6].sum()), cn[:, 1].mean(), int(sim.moves.item())))
    sim.env.close()
