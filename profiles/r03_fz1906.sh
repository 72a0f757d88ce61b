cd $GRAFT_REPO_ROOT
export VDS_FUZZ_DAYS_N=1910 VDS_FUZZ_N=1 VDS_FUZZ_MEDIUM_N=1
for i in 1 2 3; do
timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -x -k "replica_days and (1906 or 1905 or 1907)" 2>&1 | tail -60 > gpurun_out/fz1906_$i.txt
done
cat gpurun_out/fz1906_1.txt; tail -3 gpurun_out/fz1906_2.txt gpurun_out/fz1906_3.txt
