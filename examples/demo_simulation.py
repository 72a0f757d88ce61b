"""Equivalent of the reference's Demo_simulation.py (Demo_simulation.py:1-21) on the device engine.

    python examples/demo_simulation.py [data_dir]

Without an argument a small synthetic city is written to a temporary directory in the reference's ./data layout
(Node.csv, NodeIDList.txt, AccurateMap.csv, order_20161101.csv, Drivers1101.csv) and simulated; with an argument the
given directory (the reference's own ./data with the two missing blobs supplied) is used with the reference's
default parameters."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vehicles_dispatch_simulator_amd.config.setting import *          # noqa: F401,F403  (same names as the reference)
from vehicles_dispatch_simulator_amd.simulation import Simulation


def write_synthetic_data_dir(root):
    import numpy as np
    import pandas as pd
    from vehicles_dispatch_simulator_amd import synth
    city = synth.make_city(4, N=600, mode="grid", side_m=2400)
    start, pick, dele = synth.make_orders(5, city.N, 6000)
    d = os.path.join(root, "data")
    os.makedirs(d)
    pd.DataFrame({"0": np.arange(city.N), "NodeID": city.node_id, "WayID": 0, "Longitude": city.lon, "Latitude": city.lat,
                  "RoadName": "r", "Gid": 0, "Distance": 0.0}).to_csv(os.path.join(d, "Node.csv"), index=False, float_format="%.7f")
    open(os.path.join(d, "NodeIDList.txt"), "w").write("\n".join(str(int(x)) for x in city.node_id) + "\n")
    pd.DataFrame(city.cost).to_csv(os.path.join(d, "AccurateMap.csv"), header=False, index=False)
    pd.DataFrame({"ID": np.arange(start.size), "Start_time": start, "End_time": start + 600, "PointS_Longitude": city.lon[pick],
                  "PointS_Latitude": city.lat[pick], "PointE_Longitude": city.lon[dele], "PointE_Latitude": city.lat[dele],
                  "NodeS": city.node_id[pick], "NodeE": city.node_id[dele]}).to_csv(os.path.join(d, "order_20161101.csv"), index=False)
    pd.DataFrame({"DiverID": np.arange(400), "Start_time": 0, "NodeS": city.node_id[np.arange(400) % city.N]}).to_csv(
        os.path.join(d, "Drivers1101.csv"), index=False)
    return d


if __name__ == "__main__":
    os.environ.setdefault("TZ", "UTC")
    if len(sys.argv) > 1:
        data_dir, side, vehicles = sys.argv[1], SideLengthMeter, VehiclesNumber
    else:
        data_dir, side, vehicles = write_synthetic_data_dir(tempfile.mkdtemp(prefix="vds_demo_")), 2400, 300
    EXPSIM = Simulation(
        ClusterMode=ClusterMode, DemandPredictionMode=DemandPredictionMode, DispatchMode=DispatchMode,
        VehiclesNumber=vehicles, TimePeriods=TIMESTEP, LocalRegionBound=LocalRegionBound, SideLengthMeter=side,
        VehiclesServiceMeter=VehiclesServiceMeter, NeighborCanServer=NeighborCanServer, FocusOnLocalRegion=FocusOnLocalRegion,
        DataDir=data_dir)
    EXPSIM.CreateAllInstantiate()
    EXPSIM.SimCity()
