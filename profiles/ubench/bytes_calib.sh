#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE / TCC_EA0_* against known byte counts (profiles/ubench/bytes_calib.hip) on the GPU box.
#   bash profiles/ubench/bytes_calib.sh        -> gpurun_out/bytes_calib/{known.jsonl, <pass>/..._counter_collection.csv}; then
#   python profiles/ubench/bytes_calib.py      -> profiles/ubench/bytes_calib.json (committed)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bytes_calib
rm -rf $O; mkdir -p $O
B=/tmp/bytes_calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B $GRAFT_REPO_ROOT/profiles/ubench/bytes_calib.hip || exit 1
$B > $O/known.jsonl
RP="timeout 300 rocprofv3"
$RP --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.log
$RP --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.log
$RP --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/rdreq -- $B > /dev/null 2> $O/rdreq.log
$RP --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $O/wrreq -- $B > /dev/null 2> $O/wrreq.log
$RP --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/hit -- $B > /dev/null 2> $O/hit.log
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_BUBBLE[A-Z_]*\|TCC_REQ[A-Za-z_]*" | sort -u > $O/counters_available.txt
cat $O/known.jsonl | cut -c1-160
