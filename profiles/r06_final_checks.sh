#!/bin/bash
# Round-6 checks of the final build on one box (through gpurun): every replica of the bench configurations against the oracle, the fuzz at
# volume, the whole GPU suite.
cd $GRAFT_REPO_ROOT
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())"
python profiles/full_check.py cfg2 1024 2>&1 | tail -1
python profiles/full_check.py cfg4 1024 2>&1 | tail -1
VDS_DENSE_TICK_FORMS=alt python profiles/full_check.py cfg2 1024 2>&1 | tail -1
VDS_DENSE_TICK_FORMS=alt python profiles/full_check.py cfg4 1024 2>&1 | tail -1
VDS_DENSE_DFS=0 python profiles/full_check.py cfg4 1024 2>&1 | tail -1
python profiles/full_check.py cfg4 1024 8 interleaved 2>&1 | tail -1
python profiles/full_check.py cfg2 1024 16 interleaved 2>&1 | tail -1
VDS_FUZZ_PROCS=8 VDS_FUZZ_VOLUME=40000 VDS_FUZZ_VOLUME_MEDIUM=1600 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -k volume 2>&1 | tail -1
VDS_FUZZ_N=1500 VDS_FUZZ_MEDIUM_N=150 VDS_FUZZ_DAYS_N=1500 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 -k "not volume" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -1
# the whole GPU suite once more with k_tick_dense's two forms alternating slot by slot (the one expected failure: the test of the adaptive
# switch itself, which this mode replaces)
VDS_DENSE_TICK_FORMS=alt timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
