"""N>1 path on CPU: two gloo ranks shard the replicas with no data-path collective and all-reduce
the int64[8] aggregate counters exactly once per episode (what bench.py does over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vehicles_dispatch_simulator_amd import dist as vdist
from vehicles_dispatch_simulator_amd import synth


def test_shard_partitions_exactly():
    for total in (1, 7, 8, 1024, 8192, 1000):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(Exception):
        vdist.shard(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = vdist.shard(total, world, rank)
    # every rank derives its replicas' vehicle seeds from the GLOBAL replica index
    nodes = synth.make_vehicle_nodes(100 + first, 50, 20, count)
    # stand-in for the per-rank device totals: a deterministic function of the rank's replicas
    counters = torch.zeros(8, dtype=torch.int64)
    for i in range(count):
        counters += torch.tensor([int(nodes[i].sum()), first + i, 1, 0, 0, 0, 0, 0])
    vdist.allreduce_counters(counters)
    tmax = vdist.max_over_ranks(0.5 + rank)
    dist.barrier()
    if rank == 0:
        np.save(out, np.concatenate([counters.numpy(), [int(tmax * 10)]]))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce(tmp_path):
    total, world = 7, 2
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = np.load(out)
    ref_nodes = synth.make_vehicle_nodes(100, 50, 20, total)     # single-process: replicas 0..6
    assert got[0] == int(ref_nodes.sum()) and got[1] == sum(range(total)) and got[2] == total
    assert got[8] == 15   # max over ranks of (0.5, 1.5)


def test_allreduce_is_noop_without_group():
    t = torch.arange(8, dtype=torch.int64)
    assert torch.equal(vdist.allreduce_counters(t.clone()), t)
    assert vdist.max_over_ranks(1.25) == 1.25
