export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --distinct-days 0 --no-hooked-leg --no-fallbacks-leg"
for g in 1 2 3; do for w in 1 0; do
VDS_RUN_GROUPS=$g VDS_WALK_WIDE=$w timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groups $g wide $w', round(d['value']), round(d['ms_per_step'],2))"
done; done
VDS_RUN_GROUPS=1 VDS_WALK_WIDE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wide_prof -- python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline --distinct-days 0 --no-hooked-leg --no-fallbacks-leg > /dev/null 2>&1
f=$(find gpurun_out/wide_prof -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-180; rm -rf gpurun_out/wide_prof
