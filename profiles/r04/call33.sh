#!/bin/bash
# last call of the round: the GPU suite on the final host build, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c33.txt; : > $O
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())" 2>/dev/null >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu >> $O
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
cut -c1-300 gpurun_out/r04_bench.json >> $O
cat $O
