#!/bin/bash
# second evidence call of the final build (src:58eb267c0f65): configs[1] once more with the reset kernel's counters (k_reset_staged),
# the 16-day leg with ITS kernel only (the first call's pattern was split by the shell and matched every k_tick_dense<true, ...>),
# one order stream per replica (128 days, day mode 2)
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
du -sh gpurun_out; ls gpurun_out/sum_r04_cfg2_days128 gpurun_out/sum_r04_cfg2_days16
