#!/bin/bash
# The multi-GPU bench on one 8 x MI355X node: one process per GPU under torch.distributed.run, replicas sharded by global index
# (weak scaling: --replicas is PER GPU), no data-path collective, one RCCL all-reduce of the int64[8] counters per day over xGMI.
#   bash scripts/run_8gpu.sh [N=8] [bench args...]        e.g.  bash scripts/run_8gpu.sh 8 --steps 100 --warmup 3
# BASELINE configs[2] (8192 replicas on 8 GPUs) = the default 1024 replicas per GPU; configs[4]'s 8-GPU half: --workload cfg5 --replicas 128.
# Same line as the driver's (and as tests/test_gpu_two_ranks.py::test_bench_py_itself_with_two_ranks, there with two ranks on one GPU).
set -e
cd "$(dirname "$0")/.."
N=${1:-8}; shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0      # the host driver supports dmabuf IPC only: without it RCCL fails with hipIpcGetMemHandle: invalid argument
export MASTER_ADDR=127.0.0.1
PORT=${MASTER_PORT:-29531}
for n in $( [ "$N" = "sweep" ] && echo 1 2 4 8 || echo $N ); do
  if [ "$n" = "1" ]; then
    python bench.py --gpus 1 "$@"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n "$@"
  fi
done
