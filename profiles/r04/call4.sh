#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c4.txt; : > $O
for lpr in 8 16; do for D in 1 16 128; do VDS_DENSE_LPR=$lpr timeout 300 python profiles/r04/probe_days2.py $D >> $O 2>&1; done; done
VDS_DENSE=0 timeout 300 python profiles/r04/probe_days2.py 128 >> $O 2>&1
rm -rf gpurun_out/prof_hooked; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hooked -- python profiles/r04/hooked_trace.py >> $O 2>&1
python - <<'PY' >> $O
import glob, pandas as pd
f = glob.glob("gpurun_out/prof_hooked/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
print(d[["Name", "Calls", "AverageNs", "Percentage"]].head(8).to_string())
PY
cat $O
