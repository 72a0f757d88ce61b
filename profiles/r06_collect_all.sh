#!/bin/bash
# Round-6 evidence of ONE build on ONE box (through gpurun): rocprofv3 kernel stats + PMC (incl. the read-request-size pass that gives exact
# HBM-side bytes, profiles/ubench/bytes_calib.json) and the trace of the default grouped run, for the three workloads the default bench
# line carries: the headline (configs[1]: k_tick_dense), neighbour search on the dense layout (configs[3]: k_tick_dense in stamp form +
# k_dfs_walk; + the per-tick durations of both kernels) and the stress configuration (configs[4], 128 replicas); summarised on the box
# into gpurun_out/sum_<tag>/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
sumup() {   # tag, kernels...
    local tag=$1; shift
    for k in "$@"; do python profiles/summarise.py $tag "$k" > /dev/null 2>&1; done
    mkdir -p gpurun_out/sum_$tag; cp profiles/$tag/* gpurun_out/sum_$tag/; cp gpurun_out/prof_$tag/build_id.txt gpurun_out/prof_$tag/groups_trace.txt gpurun_out/sum_$tag/ 2>/dev/null
    python profiles/pertick.py gpurun_out/prof_$tag/stats > gpurun_out/sum_$tag/pertick.txt 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
bash profiles/collect.sh r06_cfg2 2>&1 | tail -1 | cut -c1-200
sumup r06_cfg2 k_tick_dense k_reset
bash profiles/collect.sh r06_cfg4 --workload cfg4 2>&1 | tail -1 | cut -c1-200
sumup r06_cfg4 k_tick_dense k_dfs_walk
bash profiles/collect.sh r06_cfg5 --workload cfg5 --replicas 128 2>&1 | tail -1 | cut -c1-200
sumup r06_cfg5 k_tick_dense k_reset
du -sh gpurun_out
