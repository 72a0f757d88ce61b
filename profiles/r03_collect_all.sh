#!/bin/bash
# Round-3 evidence of ONE build on ONE box (run through gpurun): rocprofv3 stats + PMC of the headline (configs[1]) and of the
# hybrid neighbour-search tick (configs[3]), the opt-in lanes tick, the default bench line, in-kernel sections and knob sweep
# of the lanes tick.  Summarised on the box (profiles/summarise.py) into gpurun_out/sum_<tag>/ - the raw traces exceed what
# gpurun copies back.
cd $GRAFT_REPO_ROOT
sumup() {   # tag, kernels...
    local tag=$1; shift
    for k in "$@"; do python profiles/summarise.py $tag $k > /dev/null 2>&1; done
    mkdir -p gpurun_out/sum_$tag; cp profiles/$tag/* gpurun_out/sum_$tag/; cp gpurun_out/prof_$tag/build_id.txt gpurun_out/prof_$tag/groups_trace.txt gpurun_out/sum_$tag/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
bash profiles/collect.sh r03_cfg2 2>&1 | tail -1
bash profiles/collect_extra.sh r03_cfg2 > /dev/null 2>&1
python - <<'PY'
import pandas as pd, glob, json
out = {}
for name in ("ea", "tcc", "tlb"):
    fs = glob.glob("gpurun_out/prof_r03_cfg2/%s/*/*_counter_collection.csv" % name)
    if fs:
        df = pd.read_csv(fs[0]); k = df[df.Kernel_Name.str.contains("k_tick_rows")]
        out.update({n: float(v) for n, v in k.groupby("Counter_Name").Counter_Value.mean().items()})
json.dump(out, open("gpurun_out/r03_cfg2_pmc_extra.json", "w"), indent=1)
PY
sumup r03_cfg2 k_tick_rows k_reset_fast
bash profiles/collect.sh r03_cfg4_hybrid --workload cfg4 2>&1 | tail -1
sumup r03_cfg4_hybrid k_tick_rows k_dfs_walk
VDS_LANES_AUTO_MIN_R=32 VDS_LANES_LG=2 VDS_LANES_LOC=32 VDS_LANES_KEYS=8 bash profiles/collect.sh r03_lanes 2>&1 | tail -1
sumup r03_lanes k_tick_lanes
timeout 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python profiles/lanes_sweep.py cfg2 5 "lg=-1,loc=32,keys=16" "lg=0,loc=128,keys=64" "lg=1,loc=64,keys=16" "lg=2,loc=32,keys=16" "lg=2,loc=32,keys=8" "lg=3,loc=32,keys=8" 2>&1 | grep -v amdgpu > gpurun_out/r03_lanes_sweep.txt
VDS_LIB=libvds_prof.so python profiles/lanes_sections.py lg=2,loc=32,keys=8 2>&1 | grep -v amdgpu > gpurun_out/r03_lanes_sections.txt
VDS_LIB=libvds_prof.so python profiles/lanes_sections.py lg=0,loc=128,keys=64 2>&1 | grep -v amdgpu >> gpurun_out/r03_lanes_sections.txt
python profiles/lanes_tick_ablate.py lg=2,loc=32,keys=8 60 110 2>&1 | grep -v amdgpu > gpurun_out/r03_lanes_ablate.txt
du -sh gpurun_out
