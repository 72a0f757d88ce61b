#!/bin/bash
# static arrival slots replica-major with one order stream per replica (day mode 2) against the slot-major form (VDS_ARR_RMAJOR=0);
# the hooked-slot leg with two handles of 512 replicas on their own streams
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c26.txt; : > $O
B=$PWD/build
L=$PWD/vehicles_dispatch_simulator_amd/libvds.so
timeout 1500 python -m pytest tests/test_gpu_replica_days.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_dfs_shapes.py -x -q 2>&1 | tail -3 >> $O
timeout 600 python profiles/full_check.py cfg2 256 128 interleaved 2>&1 | tail -1 >> $O
timeout 600 python profiles/full_check.py cfg2 256 256 interleaved 2>&1 | tail -1 >> $O
python profiles/ab.py $L@VDS_ARR_RMAJOR=0 $L --days 100 --rounds 3 --distinct 128 >> $O 2>&1
python profiles/ab.py $L@VDS_ARR_RMAJOR=0 $L --days 100 --rounds 2 --distinct 1024 >> $O 2>&1
python profiles/ab.py $B/libvds_head.so $B/libvds_evalb.so $L --workload cfg4 --days 80 --rounds 2 >> $O 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-distinct-all --steps 40 > gpurun_out/r04_c26_bench.json 2> gpurun_out/r04_c26_bench.err
python - >> $O <<'PY'
import json
d = json.loads(open("gpurun_out/r04_c26_bench.json").read().strip().splitlines()[-1])
h = d.get("hooked_slot", {})
print("bench value %.4g ms_per_step %.3f" % (d["value"], d["ms_per_step"]))
for k, v in h.get("variants", {}).items():
    print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
print({k: v for k, v in h.items() if k not in ("variants", "note")})
PY
grep -v amdgpu.ids $O | tail -30
