# after the fix of the per-row-day match loop (slot 127 of a row without an order): the failing seed, then the whole fuzz (twice the
# day cases), then the GPU suite and the full-size checks of the row-mapped kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_fix_checks.txt
python -c "from vehicles_dispatch_simulator_amd import _lib; print('build', _lib.load().vds_build_id().decode())" 2>/dev/null > $O
echo "VDS_FUZZ_N=6000 VDS_FUZZ_MEDIUM_N=1500 VDS_FUZZ_DAYS_N=6000 pytest tests/test_gpu_fuzz.py:" >> $O
VDS_FUZZ_N=6000 VDS_FUZZ_MEDIUM_N=1500 VDS_FUZZ_DAYS_N=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -8 >> $O
echo "python -m pytest tests -m gpu -q:" >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 >> $O
for args in "cfg2 1024" "cfg2 1024 16 interleaved" "cfg2 1024 16 blocked"; do
  python profiles/full_check.py $args 2>&1 | grep -v amdgpu | tail -1 >> $O
done
cat $O
