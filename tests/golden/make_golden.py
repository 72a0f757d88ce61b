"""Golden-fixture generator.  Runs ONLY in the build container (needs /root/reference).

    python tests/golden/make_golden.py [--only NAME ...] [--skip-real]

Every fixture is *data*: inputs captured after the unmodified reference's
``CreateAllInstantiate`` plus the outputs of its ``SimCity`` loop
(``oracle/ref_harness.py``).  Tiny cities carry full vectors (per-order assignment,
per-tick observations, per-tick idle lists / arrival dicts in container order); the two
real-shape cases (N=4139, C=192, V=10 000, O=200 000 - BASELINE.json configs[1] and
configs[3] at one replica) carry per-tick observations, counters, per-order status/wait
and SHA-256 digests of the per-order vectors; their city is regenerated from the seed by
``vehicles_dispatch_simulator_amd.synth`` (the generator asserts the regenerated tables
equal what the reference loaded).
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from vehicles_dispatch_simulator_amd import synth  # noqa: E402
import ref_harness as rh  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


from dispatch_idiom import policy_factory  # noqa: E402  (oracle/dispatch_idiom.py: shared with the GPU shell test)


TINY = {
    # name: dict(city kwargs, orders, run kwargs)
    "tiny_grid": dict(city=dict(seed=101, N=300, mode="grid", side_m=3200), O=2500, oseed=1,
                      run=dict(V=120, seed=11, cluster_mode="Grid", side_m=3200, service_m=3200)),
    "tiny_grid_nbr_scarce": dict(city=dict(seed=102, N=280, mode="grid", side_m=3200), O=3000, oseed=2,
                                 run=dict(V=24, seed=12, cluster_mode="Grid", side_m=3200, service_m=8000, neighbor_can_server=True)),
    "tiny_kmeans": dict(city=dict(seed=103, N=320, C=12), O=2500, oseed=3,
                        run=dict(V=150, seed=13, cluster_mode="KmeansClustering", side_m=3200, service_m=3200)),
    "tiny_kmeans_dfs0": dict(city=dict(seed=104, N=300, C=12), O=2500, oseed=4,
                             run=dict(V=60, seed=14, cluster_mode="KmeansClustering", side_m=3200, service_m=3200, neighbor_can_server=True)),
    "tiny_kmeans_dfs1": dict(city=dict(seed=105, N=300, C=12), O=3000, oseed=5,
                             run=dict(V=60, seed=15, cluster_mode="SpectralClustering", side_m=3200, service_m=4800, neighbor_can_server=True)),
    "tiny_kmeans_dfs2": dict(city=dict(seed=106, N=300, C=12), O=3000, oseed=6,
                             run=dict(V=50, seed=16, cluster_mode="SpectralClustering", side_m=3200, service_m=8000, neighbor_can_server=True)),
    "tiny_empty_clusters_dfs2": dict(city=dict(seed=107, N=260, C=12), O=2500, oseed=7, empty=[3, 8],
                                     run=dict(V=40, seed=17, cluster_mode="TransportationClustering", side_m=3200, service_m=8000, neighbor_can_server=True)),
    "tiny_dispatch": dict(city=dict(seed=108, N=300, C=12), O=2500, oseed=8, dispatch=True,
                          run=dict(V=140, seed=18, cluster_mode="KmeansClustering", side_m=3200, service_m=3200)),
    "tiny_dispatch_dfs2": dict(city=dict(seed=109, N=300, C=12), O=3000, oseed=9, dispatch=True,
                               run=dict(V=70, seed=19, cluster_mode="KmeansClustering", side_m=3200, service_m=8000, neighbor_can_server=True)),
    "tiny_focus_grid": dict(city=dict(seed=111, N=420, C=12), O=3000, oseed=11, focus=(104.035, 104.105, 30.625, 30.695),
                            run=dict(V=80, seed=21, cluster_mode="Grid", side_m=2000, service_m=5000, neighbor_can_server=True)),
    "tiny_tick5": dict(city=dict(seed=112, N=300, C=12), O=2500, oseed=12,
                       run=dict(V=100, seed=22, cluster_mode="KmeansClustering", side_m=3200, service_m=8000, neighbor_can_server=True, tick_minutes=5)),
    "tiny_window6": dict(city=dict(seed=113, N=300, C=12), O=2500, oseed=13,
                         run=dict(V=140, seed=23, cluster_mode="KmeansClustering", side_m=3200, service_m=3200, pickup_window_raw=6)),
    "tiny_window4_dfs2": dict(city=dict(seed=114, N=300, C=12), O=3000, oseed=14,
                              run=dict(V=70, seed=24, cluster_mode="SpectralClustering", side_m=3200, service_m=8000, neighbor_can_server=True, pickup_window_raw=4)),
    # the dispatch body books 7 extra minutes on top of the road cost (arrival time written by the body itself)
    "tiny_dispatch_delay": dict(city=dict(seed=115, N=300, C=12), O=2500, oseed=15, dispatch=True, extra_minutes=7, nbr_table=True,
                                run=dict(V=140, seed=25, cluster_mode="KmeansClustering", side_m=3200, service_m=3200)),
    # 20 000 orders on a tiny city: ~14 per minute, so ReadOrder's unstable sort (readfiles.py:78) really permutes ties
    "tiny_sort_ties": dict(city=dict(seed=116, N=300, C=12), O=20000, oseed=16, nbr_table=True,
                           run=dict(V=150, seed=26, cluster_mode="KmeansClustering", side_m=3200, service_m=3200)),
    # the neighbour-distance table the reference computes and caches (:594-621), with two empty clusters (99999 rows)
    "tiny_nbr_empty": dict(city=dict(seed=117, N=260, C=12), O=600, oseed=17, empty=[2, 9], nbr_table=True,
                           run=dict(V=40, seed=27, cluster_mode="TransportationClustering", side_m=3200, service_m=8000, neighbor_can_server=True)),
    # FRACTIONAL road costs (one decimal): the only floating point on the path is int(float) in RoadCost
    # (simulator.py:263-264).  AccurateMap.csv holds values like 9.9 / 10.0 / 0.4 and asymmetric pairs; the reference truncates
    "tiny_fraccost": dict(city=dict(seed=118, N=300, C=12, frac=True), O=2500, oseed=18, nbr_table=True,
                          run=dict(V=140, seed=28, cluster_mode="KmeansClustering", side_m=3200, service_m=3200)),
    "tiny_fraccost_dfs2": dict(city=dict(seed=119, N=300, C=12, frac=True), O=3000, oseed=19,
                               run=dict(V=50, seed=29, cluster_mode="SpectralClustering", side_m=3200, service_m=8000, neighbor_can_server=True)),
    "tiny_two_orders": dict(city=dict(seed=110, N=120, C=12), O=2, oseed=10,
                            run=dict(V=30, seed=20, cluster_mode="KmeansClustering", side_m=3200, service_m=3200)),
}

REAL = {
    "real_kmeans192": dict(city=dict(seed=2016, N=4139, C=192), O=200000, oseed=1101,
                           run=dict(V=10000, seed=1234, cluster_mode="KmeansClustering", side_m=800, service_m=800)),
    "real_spectral192_dfs2": dict(city=dict(seed=2016, N=4139, C=192), O=200000, oseed=1101,
                                  run=dict(V=10000, seed=1234, cluster_mode="SpectralClustering", side_m=800, service_m=2000, neighbor_can_server=True)),
}


# The three clusterings the reference ships (data/*Clusters.csv + *Neighbor.csv) on its real 4139-node road graph
# (Node.csv / NodeIDList.txt).  The all-pairs cost matrix and the order file are NOT shipped
# (.MISSING_LARGE_BLOBS), so travel minutes are synthesised from the real coordinates and the day is synthetic;
# node -> cluster labels and the neighbour tables (Transportation: 18 empty clusters, 99999 rows, asymmetric
# edges) are the shipped ones, parsed by the reference itself.
SHIPPED = {
    "real_shipped_kmeans": dict(mode="KmeansClustering", seed=3001, O=200000, oseed=1102,
                                run=dict(V=10000, seed=4321, side_m=800, service_m=800)),
    "real_shipped_spectral_dfs2": dict(mode="SpectralClustering", seed=3002, O=200000, oseed=1103,
                                       run=dict(V=6000, seed=4322, side_m=800, service_m=2000, neighbor_can_server=True)),
    "real_shipped_transport_dfs2": dict(mode="TransportationClustering", seed=3003, O=200000, oseed=1104,
                                        run=dict(V=6000, seed=4323, side_m=800, service_m=2000, neighbor_can_server=True)),
}
REF_DATA = os.path.join(rh.REFERENCE_ROOT, "data")
NBR_KEEP = 96     # neighbour-distance cells kept per cluster row in the fixture (longest shipped list: 81)


def shipped_paths(mode):
    stem = os.path.join(REF_DATA, str(synth.DEFAULT_BOUND) + "192" + mode)
    return stem + "Clusters.csv", stem + "Neighbor.csv"


def build_shipped_city(spec):
    import pandas as pd
    from vehicles_dispatch_simulator_amd import world
    node = pd.read_csv(os.path.join(REF_DATA, "Node.csv"))
    node_id = world.read_node_id_list(os.path.join(REF_DATA, "NodeIDList.txt"))
    index = {int(v): i for i, v in enumerate(node_id)}
    internal = np.array([index[int(v)] for v in node["NodeID"].values], dtype=np.int64)
    N = node_id.size
    b = synth.DEFAULT_BOUND
    ix = np.zeros(N, dtype=np.int64); iy = np.zeros(N, dtype=np.int64)
    ix[internal] = np.rint((node["Longitude"].values - b[0]) * 1e6).astype(np.int64)
    iy[internal] = np.rint((node["Latitude"].values - b[2]) * 1e6).astype(np.int64)
    lab_path, nb_path = shipped_paths(spec["mode"])
    labels = pd.read_csv(lab_path).values.flatten().astype(np.int32)      # one label per Node.csv row
    n2c = np.full(N, -1, dtype=np.int32)
    n2c[internal] = labels
    city = synth.City(N=N, C=192, bound=b, ix=ix, iy=iy, node_id=node_id.astype(np.int64), node2cluster=n2c,
                      cost=synth.lattice_cost(spec["seed"], ix, iy), neighbors=world._parse_neighbor_csv(nb_path, 192))
    return city, nb_path


def neighbor_cells(nb_path, keep):
    """First `keep` "(cluster, mean cost)" cells of every row of a shipped Neighbor.csv, as arrays."""
    import ast
    import pandas as pd
    rows = pd.read_csv(nb_path, header=None).values
    ids = np.zeros((rows.shape[0], keep), dtype=np.int16)
    dist = np.zeros((rows.shape[0], keep), dtype=np.float64)
    for i, row in enumerate(rows):
        for j in range(keep):
            ids[i, j], dist[i, j] = ast.literal_eval(row[j])
        # every cell beyond `keep` is irrelevant to the parse: >= 4 entries seen and not below the threshold
        assert all(ast.literal_eval(c)[1] >= 15 for c in row[keep:])
    return ids, dist


def build_city(spec):
    city = synth.make_city(**spec["city"])
    if spec.get("empty"):
        # Transportation-shape: some cluster ids own no node (reference: 99999 sentinel, simulator.py:608-609)
        lab = city.node2cluster.copy()
        for e in spec["empty"]:
            lab[lab == e] = (e + 1) % city.C
        city.node2cluster = lab
        city.neighbors = synth.cluster_neighbors_from_cost(city.cost, lab, city.C)
    return city


def generate(name, spec, real=False, shipped=False):
    t0 = time.time()
    extra = {}
    if shipped:
        city, nb_path = build_shipped_city(spec)
        extra = dict(neighbor_csv=nb_path, capture_dfs=(0, 1, 2, 3), cluster_mode=spec["mode"])
        spec = dict(spec, city=dict(seed=spec["seed"], mode="shipped"))
    else:
        city = build_city(spec)
    start, pick, dele = synth.make_orders(spec["oseed"], city.N, spec["O"])
    run = dict(spec["run"], **extra)
    pol = policy_factory(city.N) if spec.get("dispatch") else None
    focus = spec.get("focus")
    if spec.get("extra_minutes"):
        run["dispatch_extra_minutes"] = spec["extra_minutes"]
    if spec.get("nbr_table"):
        run["capture_neighbor_table"] = True
    out = rh.run_reference(city, start, pick, dele, dispatch_policy=pol, capture_lists=not real, focus_bound=focus, **run)
    # the generator's tables must be exactly what the reference loaded / derived
    if city.cost_float is None:
        assert out["cost_is_integral"]
    else:
        # the reference read the float table back from the CSV unchanged, and its RoadCost is the truncation
        F = city.cost_float
        assert not out["cost_is_integral"] and (out["cost_float"] == F).all()
        assert (np.trunc(F) != np.rint(F)).any() and (F == 9.9).any() and (F == 10.0).any() and (F == 0.4).any()
        assert (np.trunc(F) != np.trunc(F.T)).any(), "no asymmetric pair"
        assert (out["o_value"] == np.trunc(F[out["o_delivery"], out["o_pickup"]])).all()
    assert (out["cost"] == city.cost).all()
    import random
    if focus is None:
        assert (out["node2cluster"] == city.node2cluster).all()
        nbr = [out["nbr_idx"][out["nbr_off"][c]:out["nbr_off"][c + 1]].tolist() for c in range(city.C)]
        assert nbr == [list(x) for x in city.neighbors], "neighbour lists differ from the reference's"
        assert (synth.init_vehicle_nodes(random.Random(run["seed"]), city.N, run["V"]) == out["veh_node"]).all()
        assert (out["veh_cluster"] == city.node2cluster[out["veh_node"]]).all()
    else:
        valid = out["node2cluster"] >= 0
        assert 0 < valid.sum() < city.N
        assert (synth.init_vehicle_nodes(random.Random(run["seed"]), city.N, run["V"], valid) == out["veh_node"]).all()
        assert (out["veh_cluster"] == out["node2cluster"][out["veh_node"]]).all()
    meta = dict(city_seed=np.int64(spec["city"]["seed"]), city_mode=np.str_(spec["city"].get("mode", "cluster")),
                city_side_m=np.float64(spec["city"].get("side_m", 800.0)), city_frac=np.bool_(spec["city"].get("frac", False)),
                empty=np.array(spec.get("empty", []), dtype=np.int32),
                order_seed=np.int64(spec["oseed"]), n_orders_raw=np.int64(spec["O"]),
                focus_bound=np.array(spec.get("focus", ()), dtype=np.float64), cluster_mode=np.str_(run["cluster_mode"]),
                side_m=np.float64(run["side_m"]), service_m=np.float64(run["service_m"]))
    out.update(meta)
    out["sha_status"], out["sha_vehicle"], out["sha_wait"] = np.str_(sha(out["o_status"])), np.str_(sha(out["o_vehicle"])), np.str_(sha(out["o_wait"].astype(np.int32)))
    if real:
        # city regenerated from the seed on the test side; keep orders (post ReadOrder sort) compactly
        for k in ("cost", "veh_cluster") + (() if shipped else ("node2cluster",)):
            out.pop(k)
        if shipped:
            # the shipped tables travel as data: coordinates (micro-degrees), raw node ids, labels, neighbour cells
            out["ix"], out["iy"] = city.ix.astype(np.int32), city.iy.astype(np.int32)
            out["node_id"] = city.node_id
            out["node2cluster"] = out["node2cluster"].astype(np.int16)
            out["nbr_cell_id"], out["nbr_cell_dist"] = neighbor_cells(nb_path, NBR_KEEP)
        out["o_pickup"] = out["o_pickup"].astype(np.uint16)
        out["o_delivery"] = out["o_delivery"].astype(np.uint16)
        out["o_release_min"] = out["o_release_min"].astype(np.int16)
        out["veh_node"] = out["veh_node"].astype(np.uint16)
        out["o_vehicle_head"] = out["o_vehicle"][:8192].copy()
        out.pop("o_vehicle")
        out.pop("o_value")
        out["o_wait"] = out["o_wait"].astype(np.int16)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s ticks=%d orders=%d reject=%d wait=%d dispatch=%d ref_sim=%.1fs total=%.1fs size=%.0fKB" % (
        name, out["n_ticks"], out["order_num"], out["reject_num"], out["wait_sum"], out["dispatch_num"],
        out["ref_sim_s"], time.time() - t0, os.path.getsize(path) / 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--skip-real", action="store_true")
    a = ap.parse_args()
    if not rh.reference_available():
        raise SystemExit("reference not mounted: golden fixtures can only be generated in the build container")
    for name, spec in TINY.items():
        if a.only and name not in a.only:
            continue
        generate(name, spec)
    if not a.skip_real:
        for name, spec in REAL.items():
            if a.only and name not in a.only:
                continue
            generate(name, spec, real=True)
        for name, spec in SHIPPED.items():
            if a.only and name not in a.only:
                continue
            generate(name, spec, real=True, shipped=True)


if __name__ == "__main__":
    main()
