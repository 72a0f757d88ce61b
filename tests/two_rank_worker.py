"""Worker of tests/test_gpu_two_ranks.py: one rank of a 2-rank job on ONE MI355X (both ranks use cuda:0; the collective runs
over gloo because RCCL refuses two ranks on one device).  Everything else is bench.py's multi-GPU path: replicas sharded by
global index, no data-path collective, the int64[8] counters all-reduced once per day, max-over-ranks timing."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vehicles_dispatch_simulator_amd import dist as vdist  # noqa: E402
from vehicles_dispatch_simulator_amd import workloads  # noqa: E402


def main():
    total = int(sys.argv[1])
    neighbor = sys.argv[2] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    w = workloads.tiny(neighbor=neighbor, vehicles=60 if neighbor else 150)
    first, count = vdist.shard(total, world, rank)
    init = w.vehicle_nodes(count, first_replica=first)
    env = w.make_env(count, device=0, stream=torch.cuda.current_stream().cuda_stream)
    env.reset(init)
    totals = torch.zeros(8, dtype=torch.int64, device="cuda")
    dist.barrier()
    t0 = time.perf_counter()
    env.run(env.T)
    env.reduce_counters_into(totals.data_ptr())
    torch.cuda.synchronize()
    host = totals.cpu()
    vdist.allreduce_counters(host)
    elapsed = vdist.max_over_ranks(time.perf_counter() - t0)
    env.sync()
    mine = env.counters().sum(axis=0)
    kern = env.main_kernel()
    env.close()
    if rank == 0:
        print(json.dumps({"world": world, "kernel": kern, "totals": host.tolist(), "rank0_orders": int(mine[0]), "elapsed": elapsed,
                          "allreduce_calls": vdist.ALLREDUCE_CALLS}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
