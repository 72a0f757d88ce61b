"""Summarise the raw rocprofv3 CSVs of profiles/collect.sh into profiles/<tag>/ (committed).

    python profiles/summarise.py <tag> [kernel-substring]
"""
import glob
import json
import os
import shutil
import sys

import pandas as pd

tag = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_tick_rows"
LAST = int(sys.argv[3]) if len(sys.argv) > 3 else (148 if kern.startswith(("k_tick", "k_dfs")) else 0)
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
st = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # gpurun merges runs: newest
shutil.copy(st, os.path.join(dst, "kernel_stats.csv"))
out = {"_note": "rocprofv3 --pmc, one pass per group (FETCH_SIZE | WRITE_SIZE | SQ_* ...), per-launch means for kernels matching '%s' (tick kernels: over the launches of the run's last day); "
                "FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports wide coalesced reads by up to 2x on gfx950 "
                "(MI355X_MICROARCH.md, HBM section); SQ_*_CYCLES are quad-cycles summed over waves" % kern}
for name in ("fetch", "write", "rdreq", "wrreq", "sq", "sq2"):
    fs = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    df = pd.read_csv(max(fs, key=os.path.getmtime))
    k = df[df["Kernel_Name"].str.contains(kern, regex=False)]
    # the tick kernels: the LAST day of the run only (148 launches with VDS_RUN_GROUPS=1) - the days before it are the bench's settling
    # days, in which k_tick_dense still runs one form in every slot (vds_api.hip adapt_dense); the last day is the one that is timed
    if LAST and len(k) and "Dispatch_Id" in k.columns:
        ids = sorted(k["Dispatch_Id"].unique())
        if len(ids) > LAST:
            k = k[k["Dispatch_Id"].isin(set(ids[-LAST:]))]
    g = k.groupby("Counter_Name")["Counter_Value"].agg(["mean", "count"])
    for n, row in g.iterrows():
        out[n] = {"mean_per_launch": float(row["mean"]), "launches": int(row["count"])}
    if len(k):
        # as rocprofv3 prints them: on gfx950 VGPR_Count is HALF the allocated registers (the compiler's
        # -Rpass-analysis=kernel-resource-usage says 72 for k_tick_rows<true, false>, this column 36) and
        # LDS_Block_Size shows the static part only (the kernels use dynamic LDS: 18 KB per workgroup for k_tick_rows)
        out["_dispatch"] = {c: int(k[c].iloc[0]) for c in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size")}
bid = os.path.join(src, "build_id.txt")
if os.path.exists(bid):
    out["_build"] = open(bid).read().strip()
import re
json.dump(out, open(os.path.join(dst, "pmc_%s.json" % re.sub(r"[^A-Za-z0-9_]+", "_", kern).strip("_")), "w"), indent=1)
print(open(os.path.join(dst, "kernel_stats.csv")).read()[:1500])
print(json.dumps(out, indent=1))
