"""BASELINE.json configs[4] (stress: 2048 clusters / 16 384 nodes / 100 000 vehicles / 2 000 000 synthetic orders per
replica-day) at full per-replica size, R = 8 replicas on one GPU:

  * EVERY replica against the CPU oracle, bit-exact (per-order status / vehicle / wait, counters incl. evaluations);
  * the per-tick, size-independent properties of test_gpu_full_size_properties.py (vehicle conservation,
    rejects + matched == orders, vehicles taken == orders matched, twins stay identical);
  * the day repeated after vds_reset_again.

Also here: bench.py driven through ``python -m torch.distributed.run --nproc-per-node 1`` exactly as the round-end
driver launches the multi-GPU runs, so that the process-group / RCCL all-reduce path is executed on a GPU box.
"""
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


def test_stress_config_every_replica_vs_oracle():
    R = 8
    w = workloads.stress()
    V = w.vehicles
    assert (w.city.C, w.city.N, V, w.release_min.size) == (2048, 16384, 100000, 2000000)
    init = w.vehicle_nodes(R)
    init[7] = init[2]                                # twins at different rows / wavefronts
    env = w.make_env(R)
    assert env.main_kernel() == "k_tick_dense"
    env.reset(init)
    T = env.T
    prev_matched = np.zeros(R, dtype=np.int64)
    for t in range(T):
        env.step()
        ob, cn = env.obs(), env.counters()
        idle, fly = ob["idle_now"].sum(axis=1), ob["inflight"].sum(axis=1)
        assert ((idle + fly) == V).all(), "tick %d: a vehicle is in no container or in two" % t
        assert (cn[:, 0] == cn[0, 0]).all() and (cn[:, 1] + cn[:, 2] == cn[:, 0]).all()
        taken = ob["idle_pre"].sum(axis=1) - idle
        np.testing.assert_array_equal(taken, cn[:, 2] - prev_matched, err_msg="tick %d: vehicles taken != orders matched" % t)
        prev_matched = cn[:, 2].copy()
        env.advance()
    cn = env.counters()
    got = env.orders()
    assert cn[0, 0] == w.release_min.size - 1        # quirk Q1
    for k in ("status", "vehicle", "wait"):
        np.testing.assert_array_equal(got[k][7], got[k][2])
    o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server,
               w.release_min, w.pickup, w.delivery, V)
    for r in range(R):
        o.reset(init[r]); o.run_day()
        exp, oc = o.orders(), o.counters()
        for k in ("status", "vehicle", "wait"):
            np.testing.assert_array_equal(got[k][r], exp[k], err_msg="replica %d %s" % (r, k))
        assert (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
    tot = env.total_counters()
    env.reset_again()
    env.run(T)
    assert (env.total_counters() == tot).all()
    again = env.orders(3, 1)
    np.testing.assert_array_equal(again["vehicle"][0], got["vehicle"][3])
    env.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_under_torch_distributed_run_one_rank():
    """The driver's multi-GPU launch line with one rank: init_process_group("nccl"), replica shard by global index,
    the int64[8] counter all-reduce and the max-over-ranks timing all execute over RCCL on this box."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--replicas", "128", "--no-cpu-baseline", "--check", "--no-neighbour-leg", "--distinct-days", "0", "--no-hooked-leg", "--no-fallbacks-leg", "--no-stress-leg"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = [l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert out["collective"]["backend"] == "nccl" and out["collective"]["allreduce_calls"] >= 2
    assert out["collective"]["allreduce_us"] > 0 and out["per_rank"]["first_replica"] == [0]       # (event-timed RCCL all-reduce)
    assert out["parity_check_vs_oracle"] is True
    assert out["aggregate_counters_last_day"]["order_num"] == 128 * 199999
    assert 0 < out["roofline"]["frac"] <= 1.0
