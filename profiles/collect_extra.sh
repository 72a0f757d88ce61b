#!/bin/bash
# Extra PMC passes: what the scattered ring posts cost - write-request widths and TLB behaviour.
#   bash profiles/collect_extra.sh <tag> [bench args...]
export TMPDIR=/tmp
export VDS_RUN_GROUPS=1      # one launch per tick over all replicas: per-launch counters of the kernel by itself
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-neighbour-leg --distinct-days 0 $@"
rocprofv3 -L > $O/counters_list.txt 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/ea -- $B --steps 1 --warmup 0 > $O/ea.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --kernel-trace --output-format csv -d $O/tcc -- $B --steps 1 --warmup 0 > $O/tcc.log 2>&1
timeout 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $O/tlb -- $B --steps 1 --warmup 0 > $O/tlb.log 2>&1
tail -2 $O/ea.log $O/tcc.log $O/tlb.log | cut -c1-200
