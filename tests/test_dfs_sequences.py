"""Neighbour-search visit order (FindServerVehicleFunction, simulator.py:978-996) and the cached neighbour
table parse (CreateCluster, simulator.py:626-646) on the three clusterings the reference ships.

Expected values were captured from the unmodified reference (tests/golden/make_golden.py, SHIPPED): its
`Cluster.Neighbor` lists after parsing the shipped `...Neighbor.csv`, and - for depth limits 0..3 - the insertion
order of its own `Visitlist` dict from every start cluster.  Transportation has empty clusters (99999 rows) and
asymmetric edges; Spectral has one asymmetric edge.  CPU only: `vds_dfs_sequences` is host code of the library."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from vehicles_dispatch_simulator_amd import _lib, synth, workloads, world

SHIPPED = ["real_shipped_kmeans", "real_shipped_spectral_dfs2", "real_shipped_transport_dfs2"]


def raw(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.mark.parametrize("name", SHIPPED)
@pytest.mark.parametrize("depth", [0, 1, 2, 3])
def test_visit_order_matches_reference(name, depth):
    g = raw(name)
    exp_off, exp_seq = g["dfs_off_d%d" % depth], g["dfs_seq_d%d" % depth].astype(np.int32)
    off, seq = workloads.native_dfs_sequences(g["nbr_off"], g["nbr_idx"], depth)
    np.testing.assert_array_equal(off, exp_off)
    np.testing.assert_array_equal(seq, exp_seq)
    nbr = [g["nbr_idx"][g["nbr_off"][c]:g["nbr_off"][c + 1]].tolist() for c in range(int(g["C"]))]
    py = synth.dfs_sequences(nbr, depth)
    assert [x for s in py for x in s] == exp_seq.tolist()
    assert np.cumsum([0] + [len(s) for s in py]).tolist() == exp_off.tolist()


@pytest.mark.parametrize("name", SHIPPED)
def test_shipped_tables_have_the_documented_quirks(name):
    g = raw(name)
    Cn = int(g["C"])
    nbr = [set(g["nbr_idx"][g["nbr_off"][c]:g["nbr_off"][c + 1]].tolist()) for c in range(Cn)]
    asym = sum(1 for a in range(Cn) for b in nbr[a] if a not in nbr[b])
    sizes = np.bincount(g["node2cluster"][g["node2cluster"] >= 0], minlength=Cn)
    if "transport" in name:
        assert (sizes == 0).sum() == 18 and asym > 0
    else:
        assert (sizes == 0).sum() == 0
    assert all(len(x) >= 4 for x in nbr) and max(len(x) for x in nbr) <= 96
    # depth 2 reaches far more than the direct neighbours, but not everything (a DFS, not a BFS ball)
    d1 = np.diff(g["dfs_off_d1"]); d2 = np.diff(g["dfs_off_d2"])
    assert (d1 == np.array([len(x) for x in nbr]) + 1).all() and (d2 >= d1).all() and d2.max() <= Cn


@pytest.mark.parametrize("name", SHIPPED)
def test_cached_neighbor_file_parse_matches_reference(name, tmp_path):
    g = raw(name)
    ids, dist = g["nbr_cell_id"], g["nbr_cell_dist"]
    # the reference's own file format: one row per cluster of "(id, mean cost)" cells (pandas to_csv of tuples)
    import pandas as pd
    rows = [[(int(i), float(d)) for i, d in zip(ri, rd)] for ri, rd in zip(ids, dist)]
    path = os.path.join(str(tmp_path), "Neighbor.csv")
    pd.DataFrame(rows).to_csv(path, header=0, index=0)
    got = world._parse_neighbor_csv(path, int(g["C"]))
    exp = [g["nbr_idx"][g["nbr_off"][c]:g["nbr_off"][c + 1]].tolist() for c in range(int(g["C"]))]
    assert got == exp


def test_small_graphs_and_edge_cases():
    lib = _lib.load()
    # chain 0-1-2-3 with a shortcut 0->3: the deep branch claims 3 first at depth 3 (quirk Q4)
    nbr = [[1, 3], [2], [3], []]
    off = np.cumsum([0] + [len(x) for x in nbr]).astype(np.int32)
    idx = np.array([k for x in nbr for k in x], dtype=np.int32)
    for depth in (-1, 0, 1, 2, 3, 7):
        o, s = workloads.native_dfs_sequences(off, idx, depth)
        py = synth.dfs_sequences(nbr, depth)
        assert s.tolist() == [x for q in py for x in q] and o.tolist() == np.cumsum([0] + [len(q) for q in py]).tolist()
    o, s = workloads.native_dfs_sequences(off, idx, 3)
    assert s[o[0]:o[1]].tolist() == [0, 1, 2, 3]
    o, s = workloads.native_dfs_sequences(off, idx, 1)
    assert s[o[0]:o[1]].tolist() == [0, 1, 3]
    # self loops and duplicates are harmless; a long chain does not overflow the native stack
    n = 60000
    chain_off = np.arange(n + 1, dtype=np.int32); chain_off[-1] = n - 1
    chain_idx = np.arange(1, n, dtype=np.int32)
    seq_off = np.zeros(n + 1, dtype=np.int32)
    rc = lib.vds_dfs_sequences(chain_off.ctypes.data_as(C.c_void_p), chain_idx.ctypes.data_as(C.c_void_p), n, 2,
                               seq_off.ctypes.data_as(C.c_void_p), None, 0)
    assert rc == -3 and seq_off[n] == 3 * n - 3      # VDS_ECAPACITY with the size reported
    # one start-to-end walk of a 20 000-cluster chain per start cluster: 2e8 visits, 20 000 levels deep
    n = 20000
    chain_off = np.arange(n + 1, dtype=np.int32); chain_off[-1] = n - 1
    seq_off = np.zeros(n + 1, dtype=np.int32)
    rc = lib.vds_dfs_sequences(chain_off.ctypes.data_as(C.c_void_p), chain_idx.ctypes.data_as(C.c_void_p), n, n,
                               seq_off.ctypes.data_as(C.c_void_p), None, 0)
    assert rc == -3 and seq_off[n] == n * (n + 1) // 2
    # bad arguments
    bad = np.array([5], dtype=np.int32)
    o1 = np.array([0, 1], dtype=np.int32)
    assert lib.vds_dfs_sequences(o1.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p), 1, 1,
                                 seq_off.ctypes.data_as(C.c_void_p), None, 0) == -1
    assert lib.vds_dfs_sequences(None, None, 1, 1, seq_off.ctypes.data_as(C.c_void_p), None, 0) == -1
