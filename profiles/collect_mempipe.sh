#!/bin/bash
# (every pass under its own timeout: a derived counter - TA_BUSY_avr - hung rocprofv3 for the whole call once)
# PMC passes on the vector-memory pipeline of the headline kernel (TA / TCP busy and stall cycles, request latencies):
# is k_tick_rows bound by the per-CU address / tag pipeline rather than by VALU issue or HBM bytes?
#   bash profiles/collect_mempipe.sh <out.json> [bench args...]
export TMPDIR=/tmp
OUT=$1; shift
O=/tmp/mempipe; rm -rf $O; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-neighbour-leg --distinct-days 0 $@ --steps 1 --warmup 0"
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 100 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- $B > $O/p$i.log 2>&1)
done
python - "$OUT" <<'PY'
import pandas as pd, glob, json, sys
out = {}
for f in glob.glob("/tmp/mempipe/p*/*/*_counter_collection.csv"):
    df = pd.read_csv(f)
    k = df[df.Kernel_Name.str.contains("k_tick_rows")]
    out.update({n: float(v) for n, v in k.groupby("Counter_Name").Counter_Value.mean().items()})
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
