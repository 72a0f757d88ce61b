cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_run_hooked.py -x -q 2>&1 | tail -4
for f in 1 0; do
VDS_HOOK_FUSE=$f timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-fallbacks-leg > gpurun_out/b_hook_f$f.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/b_hook_f$f.json").read().strip().splitlines()[-1])
h=d["hooked_slot"]
print("fuse $f", d["value"], d["ms_per_step"], h.get("INVALID"))
for k,v in h["variants"].items(): print("   ", k, round(v["slot_us"],1), round(v["host_issue_us_per_slot"],1), v["dispatches_last_day"])
print("   ", {k:round(v,3) for k,v in h.items() if k.endswith("hookless_tick")})
PY
done
