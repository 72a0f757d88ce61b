"""Shared helpers for the parity tests: fixture loading and oracle replay."""
from __future__ import annotations

import glob
import os
import random

import numpy as np

from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix="tiny_"):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    """Return the fixture as a dict with every input in full int32 form.

    Real-shape fixtures do not store the city: it is regenerated from the recorded seed
    (the generator asserted equality with what the reference loaded)."""
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    if "cost" not in g and str(g["city_mode"]) == "shipped":
        # the reference's shipped road graph and clustering (labels / neighbour lists are in the fixture);
        # its cost matrix is not shipped: synthesised from the real coordinates, as the generator did
        g["cost"] = synth.lattice_cost(int(g["city_seed"]), g["ix"], g["iy"])
        g["node2cluster"] = g["node2cluster"].astype(np.int32)
        g["o_value"] = g["cost"][g["o_delivery"].astype(np.int64), g["o_pickup"].astype(np.int64)].astype(np.int64)
    elif "cost" not in g:
        city = synth.make_city(int(g["city_seed"]), N=int(g["N"]), C=int(g["C"]), mode=str(g["city_mode"]),
                               side_m=float(g["city_side_m"]), with_neighbors=False)
        g["cost"] = city.cost
        g["node2cluster"] = city.node2cluster
        g["o_value"] = city.cost[g["o_delivery"].astype(np.int64), g["o_pickup"].astype(np.int64)].astype(np.int64)
    for k in ("o_pickup", "o_delivery", "o_release_min", "veh_node"):
        g[k] = g[k].astype(np.int32)
    return g


def make_oracle(g) -> Oracle:
    o = Oracle(g["cost"], g["node2cluster"], g["nbr_off"], g["nbr_idx"], int(g["depth_limit"]),
               bool(g["neighbor_can_server"]), g["o_release_min"], g["o_pickup"], g["o_delivery"], int(g["V"]), **engine_settings(g))
    o.reset(g["veh_node"])
    return o


def engine_settings(g):
    """tick length / raw pickup window of a fixture (older fixtures: the reference's defaults)."""
    return dict(tick_minutes=int(g["tick_minutes"]) if "tick_minutes" in g else 10,
                reject_threshold=int(g["reject_threshold"]) if "reject_threshold" in g else 600_000_000_000)


def dispatch_by_tick(g):
    log = g["dispatch_log"]
    out = {}
    for row in log:
        out.setdefault(int(row[0]), []).append(row)
    return out
