"""the neighbour-matrix precompute of CreateCluster (simulator.py:594-646) as a HIP segmented
reduction (vds_cluster_cost_sums): integer pair sums bit-exact against numpy, the derived table against the
``...Neighbor.csv`` the unmodified reference computed and wrote itself (fixtures ``nbr_table_*``; float tolerance 1e-6 as
north_star states - the values are in fact identical), incl. empty clusters (99999 sentinel), and the loader end to end
without a cache file."""
import os
import time

import numpy as np
import pytest

from helpers import load_golden
from oracle.ref_harness import write_reference_data_dir
from test_world_loader import rebuild_inputs
from vehicles_dispatch_simulator_amd import synth, world, workloads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_dispatch_delay", "tiny_sort_ties", "tiny_nbr_empty"])
def test_table_equals_reference_written_neighbor_csv(name, tmp_path):
    g = load_golden(name)
    C = int(g["C"])
    sums, sizes = synth.cluster_cost_sums_gpu(g["cost"], g["node2cluster"], C)
    hs, hz = synth.cluster_cost_sums_host(g["cost"], g["node2cluster"], C)
    np.testing.assert_array_equal(sums, hs)
    np.testing.assert_array_equal(sizes, hz)
    table = synth.neighbor_table_from_sums(sums, sizes)
    np.testing.assert_array_equal(np.array([[j for j, _ in row] for row in table]), g["nbr_table_id"])
    np.testing.assert_allclose(np.array([[d for _, d in row] for row in table]), g["nbr_table_dist"], rtol=0, atol=1e-6)
    if name == "tiny_nbr_empty":
        assert (sizes == 0).sum() == 2 and (g["nbr_table_dist"] == 99999).sum() == 2 * (C - 1) + 2 * (C - 2)
    nbr = [g["nbr_idx"][g["nbr_off"][c]:g["nbr_off"][c + 1]].tolist() for c in range(C)]
    assert synth.neighbors_from_table(table) == nbr
    # the loader, no cache file present: computes on the GPU, writes the cache the way the reference does (:616-621)
    city, start, pick, dele = rebuild_inputs(g)
    os.environ["TZ"] = "UTC"; time.tzset()
    write_reference_data_dir(str(tmp_path), city, start, pick, dele, n_drivers=int(g["V"]), cluster_mode=str(g["cluster_mode"]))
    data = os.path.join(str(tmp_path), "data")
    cache = os.path.join(data, str(tuple(city.bound)) + str(C) + str(g["cluster_mode"]) + "Neighbor.csv")
    assert not os.path.exists(cache)
    W = world.load_world(data, cluster_mode=str(g["cluster_mode"]), local_region_bound=synth.DEFAULT_BOUND,
                         side_length_meter=float(g["side_m"]), vehicles_service_meter=float(g["service_m"]))
    assert [list(x) for x in W.neighbors] == nbr
    assert os.path.exists(cache)
    assert world._parse_neighbor_csv(cache, C) == nbr          # second run: the cached table gives the same lists


def test_real_shape_and_ragged_cities():
    """configs[1]'s city (4139 nodes, 192 clusters of 7-53 nodes) and awkward shapes: N not a multiple of the tile,
    nodes outside every cluster, one huge cluster, costs up to 2^23."""
    w = workloads.didi_day("cfg2")
    sums, sizes = synth.cluster_cost_sums_gpu(w.city.cost, w.city.node2cluster, w.city.C)
    hs, hz = synth.cluster_cost_sums_host(w.city.cost, w.city.node2cluster, w.city.C)
    np.testing.assert_array_equal(sums, hs)
    np.testing.assert_array_equal(sizes, hz)
    rng = np.random.default_rng(5)
    for N, C in ((1, 1), (257, 3), (1000, 300), (777, 40)):
        cost = rng.integers(0, 1 << 23, size=(N, N), dtype=np.int64).astype(np.int32)
        n2c = rng.integers(-1, C, size=N).astype(np.int32)
        if N > 500:
            n2c[: N // 2] = 0                       # one cluster holding half the city
        s, z = synth.cluster_cost_sums_gpu(cost, n2c, C)
        hs, hz = synth.cluster_cost_sums_host(cost.astype(np.int64), n2c, C)
        # the host statement goes through float64 BLAS: exact below 2^53 only - recompute its big entries in integers
        exact = np.zeros((C, C), dtype=np.int64)
        members = [np.flatnonzero(n2c == c) for c in range(C)]
        for i in range(C):
            if members[i].size:
                colsum = cost[:, members[i]].astype(np.int64).sum(axis=1)
                for j in range(C):
                    exact[i, j] = colsum[members[j]].sum()
        np.testing.assert_array_equal(s, exact)
        np.testing.assert_array_equal(z, hz)
    with pytest.raises(Exception, match="out of range"):
        synth.cluster_cost_sums_gpu(np.zeros((4, 4), np.int32), np.array([0, 1, 2, 9], np.int32), 3)
