#!/bin/bash
# Round 3, after profiles/r03_collect_all.sh + refresh_side_data.py (same build): the default bench line (roofline fraction from the
# same-build PMC traffic), kernel traces of the DEFAULT grouped vds_run (overlap of the group launches), the life-cycle probe of the
# day graph, then the final checks.  Through gpurun; everything lands in gpurun_out/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --steps 2 --warmup 1"
for wl in cfg2 cfg4; do
  rm -rf /tmp/gt_$wl
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt_$wl -- $B --workload $wl > /tmp/gt_$wl.log 2>&1)
  python profiles/run_groups_trace.py $(find /tmp/gt_$wl -name "*kernel_trace.csv" | head -1) > gpurun_out/groups_trace_$wl.txt 2>&1
done
{
  for cfg in "3 2 2" "2 1 3" "3 1 3"; do
    n=0; for i in 1 2 3 4; do timeout 200 python profiles/r03_run_groups/crash_probe.py $cfg 10 > /tmp/p.txt 2>&1 || n=$((n+1)); done
    echo "crash_probe.py $cfg 10, 4 runs: $n failures"
  done
  python profiles/run_groups_sweep.py 1024 40 "1:0 2:1 1:0 2:1" cfg2 2>&1 | grep -v amdgpu | cut -c1-100
  python profiles/run_groups_sweep.py 1024 15 "1:0 3:1 1:0 3:1" cfg4 2>&1 | grep -v amdgpu | cut -c1-100
} > gpurun_out/r03_run_groups_final.txt 2>&1
bash profiles/r03_final_checks.sh > /dev/null 2>&1
cat gpurun_out/r03_final_checks.txt gpurun_out/r03_run_groups_final.txt gpurun_out/groups_trace_cfg2.txt gpurun_out/groups_trace_cfg4.txt
