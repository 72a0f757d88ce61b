"""Named synthetic workloads of BASELINE.json (`configs`), built from ``synth`` only - nothing
here reads the reference tree, so they regenerate identically on the GPU box."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import synth
from .env import neighbors_to_csr


@dataclass
class Workload:
    name: str
    city: synth.City
    release_min: np.ndarray
    pickup: np.ndarray
    delivery: np.ndarray
    vehicles: int
    neighbor_can_server: bool
    depth_limit: int
    nbr_off: np.ndarray
    nbr_idx: np.ndarray
    veh_seed: int

    def vehicle_nodes(self, replicas: int, first_replica: int = 0) -> np.ndarray:
        """[replicas, V]; replica r (global index) draws from random.Random(veh_seed + r)."""
        valid = self.city.node2cluster >= 0
        return native_vehicle_nodes(self.veh_seed + first_replica, self.city.N, self.vehicles, replicas,
                                    None if valid.all() else valid)

    def make_env(self, replicas: int, device: int = 0, stream: Optional[int] = None, load: bool = True, **kw):
        from .env import BatchedDispatchEnv
        env = BatchedDispatchEnv(self.city.cost, self.city.node2cluster, self.nbr_off, self.nbr_idx, replicas=replicas,
                                 vehicles=self.vehicles, depth_limit=self.depth_limit,
                                 neighbor_can_server=self.neighbor_can_server, device=device, stream=stream, **kw)
        if load:
            env.load_orders(self.release_min, self.pickup, self.delivery)
        return env


def native_vehicle_nodes(seed: int, N: int, V: int, R: int, valid=None) -> np.ndarray:
    """``synth.make_vehicle_nodes`` through the native MT19937 of libvds (``vds_py_random_nodes``): the same
    ``random.Random(seed + r).choice(range(N))`` streams, ~100x faster (tests/test_py_random.py pins the equality)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    out = np.empty((R, V), dtype=np.int32)
    vmask = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
    for r in range(R):
        rc = lib.vds_py_random_nodes(C.c_uint64(seed + r), N, V, None if vmask is None else vmask.ctypes.data_as(C.c_void_p),
                                     out[r].ctypes.data_as(C.c_void_p))
        if rc:
            raise Exception("vds_py_random_nodes failed (%d)" % rc)
    return out


def native_dfs_sequences(nbr_off, nbr_idx, depth_limit: int):
    """Visit order of ``FindServerVehicleFunction`` (``simulator.py:978-996``) from every start cluster, as the
    library derives it (``vds_dfs_sequences``; host code, no GPU): ``(seq_off[C+1], seq)``, start cluster first."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    off = np.ascontiguousarray(nbr_off, dtype=np.int32)
    idx = np.ascontiguousarray(nbr_idx, dtype=np.int32)
    n = off.size - 1
    seq_off = np.zeros(n + 1, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
    rc = lib.vds_dfs_sequences(p(off), p(idx), n, int(depth_limit), p(seq_off), None, 0)
    if rc not in (0, -3):
        raise Exception("vds_dfs_sequences failed (%d)" % rc)
    seq = np.zeros(int(seq_off[n]), dtype=np.int32)
    rc = lib.vds_dfs_sequences(p(off), p(idx), n, int(depth_limit), p(seq_off), p(seq), seq.size)
    if rc:
        raise Exception("vds_dfs_sequences failed (%d)" % rc)
    return seq_off, seq


def depth_limit_for(side_m: float, service_m: float) -> int:
    """NeighborServerDeepLimit, ``simulator.py:290``."""
    return int((service_m - (0.5 * side_m)) // side_m)


def didi_day(name: str = "cfg2", *, N: int = 4139, C: int = 192, vehicles: int = 10000, orders: int = 200000,
             neighbor: bool = False, service_m: float = 800.0, side_m: float = 800.0,
             city_seed: int = 2016, order_seed: int = 1101, veh_seed: int = 1234) -> Workload:
    """BASELINE.json configs[1] (``neighbor=False``) / configs[3] (``neighbor=True, service_m=2000``):
    192 k-means-shaped clusters over 4,139 nodes, 10k vehicles, ~200k orders in one day."""
    city = synth.make_city(city_seed, N=N, C=C, with_neighbors=neighbor)
    start, pick, dele = synth.make_orders(order_seed, N, orders)
    rel = synth.release_minutes(start)
    rel = (rel - rel[0]).astype(np.int32)
    off, idx = neighbors_to_csr(city.neighbors if neighbor else [[] for _ in range(city.C)])
    return Workload(name=name, city=city, release_min=rel, pickup=pick, delivery=dele, vehicles=vehicles,
                    neighbor_can_server=neighbor, depth_limit=depth_limit_for(side_m, service_m) if neighbor else 0,
                    nbr_off=off, nbr_idx=idx, veh_seed=veh_seed)


def stress(name: str = "cfg5", *, N: int = 16384, C: int = 2048, vehicles: int = 100000, orders: int = 2000000,
           city_seed: int = 2048, order_seed: int = 55, veh_seed: int = 99) -> Workload:
    """BASELINE.json configs[4] per replica: 2048 clusters, 100k vehicles, 2M synthetic orders/day.  N = 16384
    nodes (8 per cluster on average; the N x N int32 cost matrix is 1.07 GB), no neighbour search."""
    return didi_day(name, N=N, C=C, vehicles=vehicles, orders=orders, city_seed=city_seed, order_seed=order_seed, veh_seed=veh_seed)


def tiny(name: str = "tiny", *, N: int = 300, C: int = 12, vehicles: int = 150, orders: int = 2500,
         neighbor: bool = False, depth_limit: int = 2, seed: int = 7) -> Workload:
    city = synth.make_city(seed, N=N, C=C, with_neighbors=True)
    start, pick, dele = synth.make_orders(seed + 1, N, orders)
    rel = synth.release_minutes(start)
    rel = (rel - rel[0]).astype(np.int32)
    off, idx = neighbors_to_csr(city.neighbors)
    return Workload(name=name, city=city, release_min=rel, pickup=pick, delivery=dele, vehicles=vehicles,
                    neighbor_can_server=neighbor, depth_limit=depth_limit if neighbor else 0, nbr_off=off, nbr_idx=idx,
                    veh_seed=seed + 2)


# Calibration of the CPU baseline's kind "port" (bench.py `cpu_baseline`): the C oracle against the REFERENCE ITSELF on identical inputs,
# measured in the build container (the reference cannot travel to the GPU box).  Recipe: tests/golden/<fixture>.npz holds `ref_sim_s`, the
# wall time of the unmodified reference's SimCity() on that day (tests/golden/make_golden.py, one core); the oracle replays the same day
# (`helpers.make_oracle(g); o.reset(g["veh_node"]); o.run_day()`, median of 5).  reference_s / oracle_s:
REFERENCE_CALIBRATION = {
    "cfg2": {"fixture": "real_kmeans192", "reference_sim_s": 41.67, "oracle_s": 0.0407, "reference_ratio": 1024.0},
    "cfg4": {"fixture": "real_spectral192_dfs2", "reference_sim_s": 82.59, "oracle_s": 0.1168, "reference_ratio": 707.0},
    "where": "build container (one core of its host CPU), round 5; the ratio, not the absolute times, is what carries to another host",
}


def algorithmic_bytes(work: dict, vehicles: int) -> int:
    """SURVEY.md 8(d) byte model of the tick path: 16 B per (order, candidate) evaluation
    (idle entry 8 + vehicle location 4 + cost 4), 24 B per processed order, 24 B per match,
    4 B per vehicle per tick (arrival scan) and 28 B per arrival."""
    return (16 * work["evals"] + 24 * work["orders"] + 24 * work["matches"] + 4 * vehicles * work["ticks"] + 28 * work["arrivals"])


def layout_bytes(work: dict, *, replicas: int, clusters: int, idle_loaded: int, busy_buckets: int) -> int:
    """Bytes the tick kernel's DATA LAYOUT (csrc/vds_device.h) has to move through HBM for the work in ``work`` - the
    lower bound that is specific to this implementation, as opposed to ``algorithmic_bytes``:

      every (replica, cluster) bucket, every tick   header 12 B + arrival-slot counter 4 B + counters 64 B read,
                                                    header 12 B written
      buckets with orders (``busy_buckets``)        5 counter words (40 B) written
      idle entries loaded (``idle_loaded``)         8 B each ({vehicle, node}); lists of order-less buckets are not read
      arrivals                                      16 B entry read + 8 B idle entry appended
      processed orders                              1 B (16 B record shared by the 16 replicas of a workgroup) + 8 B result
      matches                                       16 B arrival entry + 8 B slot-counter read-modify-write

    Cost lookups are served from the LDS copy of the cluster block and compaction rewrites are not counted (their
    minimum is zero), so measured traffic lies above this figure."""
    buckets = work["ticks"] * clusters              # work["ticks"] = ticks x replicas
    return (buckets * (12 + 4 + 64 + 12) + 40 * busy_buckets + 8 * idle_loaded + 24 * work["arrivals"]
            + 9 * work["orders"] + 24 * work["matches"])


def profile_side_data(root: str, workload: str, replicas: int, kernel: str) -> dict:
    """Committed rocprofv3 results for (workload, replicas, kernel): PMC-measured HBM bytes per launch
    (profiles/traffic.json) and the VALU-issue limiter inputs (profiles/limiter.json).  Empty when none matches."""
    import json
    import os
    out = {}
    for fn, key in (("traffic.json", None), ("limiter.json", "limiter")):
        path = os.path.join(root, "profiles", fn)
        if not os.path.exists(path):
            continue
        try:
            for e in json.load(open(path)).get("entries", []):
                if e.get("workload") == workload and e.get("replicas") == replicas and e.get("kernel") == kernel:
                    if key is None:
                        out["hbm_bytes_per_launch"] = e.get("hbm_bytes_per_launch")
                        out["traffic_build"] = e.get("build")
                        out["traffic_basis"] = e.get("read_bytes_basis", "FETCH_SIZE x 2 (upper bound)") + " + WRITE_SIZE"
                        out["traffic_source"] = e.get("source")
                    else:
                        out[key] = e
        except Exception:
            pass
    return out


def distinct_days(w: Workload, n_days: int, order_seed: int = 7001):
    """``n_days`` different synthetic days of the workload's shape (same city, same order count, other seeds): what R
    independent cities replay when each has its own ``Orders`` (``simulator.py:325-342``).  Day 0 is the workload's own."""
    days = [(w.release_min, w.pickup, w.delivery)]
    for d in range(1, n_days):
        start, pick, dele = synth.make_orders(order_seed + d, w.city.N, w.release_min.size)
        rel = synth.release_minutes(start)
        days.append(((rel - rel[0]).astype(np.int32), pick, dele))
    return days
