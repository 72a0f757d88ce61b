export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_lanes_a
rm -rf $O; mkdir -p $O
export VDS_LANES_AUTO_MIN_R=32 VDS_LANES_LG=2
B="python bench.py --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_GDS --kernel-trace --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --kernel-trace --output-format csv -d $O/tcc -- $B > $O/tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O/write.log 2>&1
python - <<PY
import pandas as pd, glob
for name in ("sq","sq2","sq3","tcc","fetch","write"):
    fs = glob.glob("$O/%s/*/*_counter_collection.csv"%name)
    if not fs: print(name,"none"); continue
    df = pd.read_csv(fs[0]); k = df[df.Kernel_Name.str.contains("k_tick_lanes")]
    print(name, {a: round(b) for a,b in k.groupby("Counter_Name").Counter_Value.mean().items()}, len(k))
PY
