#!/bin/bash
# rocprofv3 stats of the round-4 kernels off the bench's timed path: the start-node generator (k_py_random_nodes), the device-side
# checks of uploaded start nodes (k_start_nodes_check), k_reset_staged, at configs[1] with 1024 replicas
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04_aux; rm -rf $O; mkdir -p $O
cat > $O/run.py <<'PY'
import numpy as np
from vehicles_dispatch_simulator_amd import workloads
w = workloads.didi_day("cfg2")
R = 1024
env = w.make_env(R)
init = w.vehicle_nodes(R)
seeds = (np.arange(R) + w.veh_seed).astype(np.uint64)
for _ in range(6):
    env.reset(init)
    env.reset_random(seeds)
    env.reset_again()
env.sync()
PY
PYTHONPATH=$PWD rocprofv3 --kernel-trace --stats --output-format csv -d $O/reset -- python $O/run.py > $O/reset.log 2>&1
python - <<'PY'
import glob, pandas as pd
f = glob.glob("gpurun_out/prof_r04_aux/reset/*/*_kernel_stats.csv")[0]
df = pd.read_csv(f)
df = df[df.Name.str.contains("k_")]
df.to_csv("gpurun_out/r04_aux_reset_kernel_stats.csv", index=False)
print(df[["Name", "Calls", "AverageNs", "Percentage"]].to_string())
PY
rm -rf $O
