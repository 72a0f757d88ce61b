#!/bin/bash
# Round-5 final checks of ONE build on ONE box (through gpurun); output: gpurun_out/r05_final_checks.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final_checks.txt
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())" 2>/dev/null > $O
echo "VDS_FUZZ_N=6000 VDS_FUZZ_MEDIUM_N=1500 VDS_FUZZ_DAYS_N=6000 pytest tests/test_gpu_fuzz.py:" >> $O
VDS_FUZZ_N=6000 VDS_FUZZ_MEDIUM_N=1500 VDS_FUZZ_DAYS_N=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2 >> $O
# every replica of every bench configuration against the oracle: shared day (8 lanes per replica), the same on 16 lanes / 256-entry
# tables, 16 days both maps (day mode 1, 16-row workgroups), 128 days (8 replicas per day: 8-row workgroups), 256 days (4-row workgroups),
# 1024 days (one order stream per row: day mode 2), the wide layout, neighbour search with the serial walk and with deferred acceptance
for args in "cfg2 1024" "cfg2 1024 16 interleaved" "cfg2 1024 16 blocked" "cfg2 1024 128 interleaved" "cfg2 1024 256 interleaved" "cfg2 1024 1024 interleaved" "cfg2 1024 1 interleaved 5" "cfg4 1024" "cfg4 1024 8 interleaved"; do
  timeout 900 python profiles/full_check.py $args 2>&1 | grep -v amdgpu | tail -1 >> $O
done
VDS_WALK_DA=1 timeout 900 python profiles/full_check.py cfg4 1024 2>&1 | grep -v amdgpu | tail -1 | sed 's/^/VDS_WALK_DA=1: /' >> $O
VDS_DENSE_LPR=16 timeout 900 python profiles/full_check.py cfg2 1024 2>&1 | grep -v amdgpu | tail -1 | sed 's/^/VDS_DENSE_LPR=16: /' >> $O
timeout 600 python profiles/soak.py 40 cfg2 2>&1 | grep -v amdgpu | tail -1 >> $O
timeout 600 python profiles/soak.py 12 cfg4 2>&1 | grep -v amdgpu | tail -1 >> $O
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_run_groups.py -q -k life_cycle 2>&1 | tail -1 >> $O; done
cat $O
