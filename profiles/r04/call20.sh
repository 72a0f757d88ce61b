#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c20.txt; : > $O
rm -rf gpurun_out/prof_hooked; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hooked -- python profiles/r04/hooked_trace.py > /dev/null 2>&1
python - <<'PY' >> $O
import glob, pandas as pd
f = glob.glob("gpurun_out/prof_hooked/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
for _, r in d.head(6).iterrows():
    print("%-90s calls %5d avg %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], r["AverageNs"]/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_hooked
cat $O
