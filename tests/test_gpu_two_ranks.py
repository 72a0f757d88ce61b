"""Two ranks of the real engine (libvds on the GPU, one process each) under torch.distributed.run: replica sharding by global
index, per-rank device counters, one all-reduce per day - the N > 1 code path of bench.py with the engine in it, on the one GPU
the test box has (collective over gloo: RCCL does not take two ranks on one device).  The all-reduced totals must equal the
CPU oracle's sums over ALL replicas of the job."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("neighbor", [False, True])
def test_two_ranks_share_the_job(neighbor):
    total = 11                                   # 6 + 5 replicas
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "two_rank_worker.py"), str(total), "1" if neighbor else "0"]
    res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["world"] == 2 and out["allreduce_calls"] == 1
    assert out["kernel"] == ("k_dfs_hybrid" if neighbor else "k_tick_dense")
    w = workloads.tiny(neighbor=neighbor, vehicles=60 if neighbor else 150)
    init = w.vehicle_nodes(total)
    exp = np.zeros(8, dtype=np.int64)
    first_rank_orders = 0
    for r in range(total):
        o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server,
                   w.release_min, w.pickup, w.delivery, w.vehicles)
        o.reset(init[r]); o.run_day()
        oc = o.counters()
        exp[0] += oc["order_num"]; exp[1] += oc["reject_num"]; exp[2] += oc["wait_sum"]; exp[4] += oc["evals"]
        if r < 6:
            first_rank_orders += oc["order_num"]
    got = np.array(out["totals"], dtype=np.int64)
    assert got[0] == exp[0] and got[1] == exp[1] and got[2] == exp[2] and got[4] == exp[4], (got.tolist(), exp.tolist())
    assert out["rank0_orders"] == first_rank_orders
