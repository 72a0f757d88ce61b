#!/bin/bash
# Round-4 evidence of ONE build on ONE box (through gpurun): rocprofv3 kernel stats + PMC (incl. the read-request-size pass that
# gives exact HBM-side bytes, profiles/ubench/bytes_calib.json) of the headline (configs[1]: k_tick_dense, 8 lanes per replica), of
# configs[1] with 16 order days (16 lanes per replica, 256-entry tables) and with one order stream per replica (128 days: day mode 2), of the hybrid neighbour-search tick (configs[3]) and of
# the stress configuration (configs[4], 128 replicas), summarised on the box into gpurun_out/sum_<tag>/ (the raw traces exceed
# what gpurun copies back); then the section attribution and the launch spans (in-flight histogram) of the instrumented build.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
sumup() {   # tag, kernels...
    local tag=$1; shift
    for k in "$@"; do python profiles/summarise.py $tag "$k" > /dev/null 2>&1; done
    mkdir -p gpurun_out/sum_$tag; cp profiles/$tag/* gpurun_out/sum_$tag/; cp gpurun_out/prof_$tag/build_id.txt gpurun_out/prof_$tag/groups_trace.txt gpurun_out/sum_$tag/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
bash profiles/collect.sh r04_cfg2 2>&1 | tail -1 | cut -c1-200
sumup r04_cfg2 k_tick_dense k_reset
DISTINCT=16 bash profiles/collect.sh r04_cfg2_days16 2>&1 | tail -1 | cut -c1-200
sumup r04_cfg2_days16 "k_tick_dense<true, 1"
DISTINCT=128 bash profiles/collect.sh r04_cfg2_days128 2>&1 | tail -1 | cut -c1-200
sumup r04_cfg2_days128 "k_tick_dense<true, 2"
bash profiles/collect.sh r04_cfg4_hybrid --workload cfg4 2>&1 | tail -1 | cut -c1-200
sumup r04_cfg4_hybrid k_tick_rows k_dfs_walk
VDS_DENSE_LPR=16 bash profiles/collect.sh r04_cfg5 --workload cfg5 --replicas 128 2>&1 | tail -1 | cut -c1-200
sumup r04_cfg5 k_tick_dense k_reset
if [ -f build/libvds_prof.so ]; then
  VDS_LIB=$PWD/build/libvds_prof.so VDS_RUN_GRAPH=0 timeout 300 python profiles/r04/sections_dense.py 1024 1 2>&1 | grep -v amdgpu > gpurun_out/r04_sections_dense.txt
  VDS_LIB=$PWD/build/libvds_prof.so timeout 300 python profiles/r04/inflight.py 2>&1 | grep -v amdgpu > gpurun_out/r04_inflight.txt
  VDS_LIB=$PWD/build/libvds_prof.so timeout 300 python profiles/r04/ablate_dense.py 2>&1 | grep -v amdgpu > gpurun_out/r04_ablate_dense.txt
fi
du -sh gpurun_out; cat gpurun_out/r04_inflight.txt
