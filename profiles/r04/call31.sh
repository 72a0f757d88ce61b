#!/bin/bash
# the driver's round-end sequence on the final build (smoke, bench N = 1 as the driver launches it, bench under torch.distributed.run
# with one rank), then a long fuzz
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c31.txt; : > $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu >> $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 2>/dev/null | cut -c1-330 >> $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-neighbour-leg --no-hooked-leg --distinct-days 0 --no-distinct-all 2>/dev/null | cut -c1-330 >> $O
echo "VDS_FUZZ_N=20000 VDS_FUZZ_MEDIUM_N=4000 VDS_FUZZ_DAYS_N=20000 pytest tests/test_gpu_fuzz.py:" >> $O
VDS_FUZZ_N=20000 VDS_FUZZ_MEDIUM_N=4000 VDS_FUZZ_DAYS_N=20000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2 >> $O
cat $O
