#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
#   bash profiles/collect.sh <tag> [bench args...]        (DISTINCT=16 ... : also run the per-replica-days leg)
# Writes raw CSVs under gpurun_out/prof_<tag>/ ; summarise with profiles/summarise.py, then profiles/refresh_side_data.py
export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
python -c "from vehicles_dispatch_simulator_amd import _lib; print(_lib.load().vds_build_id().decode())" > $O/build_id.txt 2>/dev/null
B="python bench.py --no-cpu-baseline --no-neighbour-leg --no-hooked-leg --no-fallbacks-leg --no-stress-leg --no-distinct-all --distinct-days ${DISTINCT:-0} $@"
# The kernel BY ITSELF: one launch per tick over all replicas (VDS_RUN_GROUPS=1) - per-launch durations and counters mean what
# they say.  The default vds_run (replica groups as parallel branches of the day graph: overlapping half-size launches) is
# traced once more at the end (stats_groups: profiles/run_groups_trace.py turns it into durations + overlap).
export VDS_RUN_GROUPS=1
RP="timeout 300 rocprofv3"
$RP --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 2 --warmup 1 > $O/stats.log 2>&1
$RP --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1
$RP --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/write.log 2>&1
# exact HBM-side bytes (profiles/ubench/bytes_calib.json: FETCH_SIZE counts every read request as 64 B whether it moves 64 or 128):
# reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B, writes = WRITE_SIZE = 32 x (WRREQ - WRREQ_64B) + 64 x WRREQ_64B
$RP --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/rdreq -- $B --steps 1 --warmup 0 > $O/rdreq.log 2>&1
$RP --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $O/wrreq -- $B --steps 1 --warmup 0 > $O/wrreq.log 2>&1
$RP --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq -- $B --steps 1 --warmup 0 > $O/sq.log 2>&1
$RP --pmc SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- $B --steps 1 --warmup 0 > $O/sq2.log 2>&1
unset VDS_RUN_GROUPS
$RP --kernel-trace --output-format csv -d $O/stats_groups -- $B --steps 2 --warmup 1 > $O/stats_groups.log 2>&1
python profiles/run_groups_trace.py $(find $O/stats_groups -name "*kernel_trace.csv" | head -1) > $O/groups_trace.txt 2>&1
tail -1 $O/stats.log | cut -c1-300
