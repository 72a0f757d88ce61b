#!/bin/bash
# Round-5 evidence of ONE build on ONE box (through gpurun): rocprofv3 kernel stats + PMC (incl. the read-request-size pass that gives exact
# HBM-side bytes, profiles/ubench/bytes_calib.json) of the headline (configs[1]: k_tick_dense, 8 lanes per replica, 32-row workgroups),
# of configs[1] with 16 order days and with 128 order days (8 replicas per day: 8-row workgroups, round 5), of the hybrid
# neighbour-search tick (configs[3]; serial walk = default, and the deferred-acceptance form) and of the stress configuration
# (configs[4], 128 replicas), summarised on the box into gpurun_out/sum_<tag>/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
sumup() {   # tag, kernels...
    local tag=$1; shift
    for k in "$@"; do python profiles/summarise.py $tag "$k" > /dev/null 2>&1; done
    mkdir -p gpurun_out/sum_$tag; cp profiles/$tag/* gpurun_out/sum_$tag/; cp gpurun_out/prof_$tag/build_id.txt gpurun_out/prof_$tag/groups_trace.txt gpurun_out/sum_$tag/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
bash profiles/collect.sh r05_cfg2 2>&1 | tail -1 | cut -c1-200
sumup r05_cfg2 k_tick_dense k_reset
DISTINCT=16 bash profiles/collect.sh r05_cfg2_days16 2>&1 | tail -1 | cut -c1-200
sumup r05_cfg2_days16 "k_tick_dense<true, 1"
DISTINCT=128 bash profiles/collect.sh r05_cfg2_days128 2>&1 | tail -1 | cut -c1-200
sumup r05_cfg2_days128 "k_tick_dense<true, 1"
bash profiles/collect.sh r05_cfg4_hybrid --workload cfg4 2>&1 | tail -1 | cut -c1-200
sumup r05_cfg4_hybrid k_tick_rows k_dfs_walk
VDS_WALK_DA=1 bash profiles/collect.sh r05_cfg4_da --workload cfg4 2>&1 | tail -1 | cut -c1-200
sumup r05_cfg4_da k_tick_rows k_dfs_walk
bash profiles/collect.sh r05_cfg5 --workload cfg5 --replicas 128 2>&1 | tail -1 | cut -c1-200
sumup r05_cfg5 k_tick_dense k_reset
du -sh gpurun_out
