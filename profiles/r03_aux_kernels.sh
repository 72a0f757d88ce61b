#!/bin/bash
# rocprofv3 stats of the kernels off the bench's timed path (VERDICT r2 "missing" 6): k_dispatch_dense / k_pack_obs /
# k_reduce_counters in a 1024-city loop with a torch policy, k_cluster_cost_sums on the 4139-node matrix.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03_aux; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loop -- python examples/batched_dispatch_loop.py 1024 > $O/loop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nbr -- python -m pytest tests/test_gpu_neighbors.py -q -m gpu > $O/nbr.log 2>&1
python - <<'PY'
import glob, pandas as pd
for tag in ("loop", "nbr"):
    f = glob.glob("gpurun_out/prof_r03_aux/%s/*/*_kernel_stats.csv" % tag)[0]
    df = pd.read_csv(f)
    df = df[df.Name.str.contains("k_")]
    df.to_csv("gpurun_out/r03_aux_%s_kernel_stats.csv" % tag, index=False)
    print(df[["Name", "Calls", "AverageNs", "Percentage"]].to_string())
PY
rm -rf $O
