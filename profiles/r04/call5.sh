#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c5.txt; : > $O
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replica_days.py tests/test_gpu_simulation_shell.py -x -q 2>&1 | tail -5) >> $O
for lpr in 8 16; do for D in 1 16 128; do VDS_LIB=$PWD/build/libvds_prof.so VDS_DENSE_LPR=$lpr timeout 300 python profiles/r04/probe_reasons.py $D >> $O 2>&1; done; done
for D in 16 128; do timeout 300 python profiles/r04/probe_days2.py $D >> $O 2>&1; done
timeout 600 python bench.py --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-distinct-all --steps 20 > gpurun_out/r04_bench5.json 2>> $O
python - <<'PY' >> $O
import json
d = json.loads(open("gpurun_out/r04_bench5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
h = d["hooked_slot"]
print({k: v for k, v in h.items() if k not in ("variants", "note")})
for k, v in h["variants"].items(): print(k, v)
PY
grep -v amdgpu.ids $O
