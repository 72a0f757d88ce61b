#!/bin/bash
# per-tick duration of k_dfs_walk over one day, serial walk vs deferred acceptance, one chain (VDS_RUN_GROUPS=1), one box
#   bash profiles/r05/walk_per_tick.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/walk_per_tick
rm -rf $O; mkdir -p $O
export VDS_RUN_GROUPS=1
B="python bench.py --workload cfg4 --no-cpu-baseline --steps 2 --warmup 1 --distinct-days 0 --no-hooked-leg --no-fallbacks-leg"
for mode in serial da; do
  if [ $mode = da ]; then export VDS_WALK_DA=1; else unset VDS_WALK_DA; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -- $B > $O/$mode.log 2>&1
done
python - <<PY
import csv, glob, collections
res = {}
for mode in ("serial", "da"):
    f = glob.glob("$O/%s/**/*kernel_trace.csv" % mode, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_dfs_walk" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    # the last full day of 148 launches
    d = d[-148:]
    res[mode] = d
out = open("$O/per_tick.txt", "w")
tot = {m: sum(v) for m, v in res.items()}
out.write("k_dfs_walk us per tick, last day of the run: serial total %.1f us, da total %.1f us\n" % (tot["serial"], tot["da"]))
out.write("sum of min(serial, da) per tick: %.1f us\n" % sum(min(a, b) for a, b in zip(res["serial"], res["da"])))
for t, (a, b) in enumerate(zip(res["serial"], res["da"])):
    out.write("%3d %8.1f %8.1f\n" % (t, a, b))
out.close()
print(open("$O/per_tick.txt").read()[:400])
s = sorted(res["serial"], reverse=True)
print("serial: top 10%% of ticks = %.0f%% of time; top 25%% = %.0f%%" % (100 * sum(s[:15]) / sum(s), 100 * sum(s[:37]) / sum(s)))
PY
rm -rf $O/serial $O/da
