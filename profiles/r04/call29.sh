#!/bin/bash
# replica groups of vds_run on the final build (host-side setting, no kernel change): configs[3] with 2 / 3 / 4 / 6 groups, configs[4]
# at 128 replicas with 1 / 2 / 4 groups (the default takes one group below 256 replicas)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c29.txt; : > $O
L=$PWD/vehicles_dispatch_simulator_amd/libvds.so
python profiles/ab.py $L@VDS_RUN_GROUPS=2 $L@VDS_RUN_GROUPS=3 $L@VDS_RUN_GROUPS=4 $L@VDS_RUN_GROUPS=6 --workload cfg4 --days 60 --rounds 2 >> $O 2>&1
python profiles/ab.py $L@VDS_RUN_GROUPS=1 $L@VDS_RUN_GROUPS=2 $L@VDS_RUN_GROUPS=4 --workload cfg5 --replicas 128 --days 100 --rounds 2 >> $O 2>&1
python profiles/ab.py $L@VDS_RUN_GROUPS=1 $L@VDS_RUN_GROUPS=2 $L@VDS_RUN_GROUPS=3 --days 300 --rounds 2 --distinct 16 >> $O 2>&1
python profiles/ab.py $L@VDS_RUN_GROUPS=1 $L@VDS_RUN_GROUPS=2 $L@VDS_RUN_GROUPS=3 --days 100 --rounds 2 --distinct 128 >> $O 2>&1
grep -v amdgpu.ids $O | sed 's#/tmp/code/szlhl1040__Vehicles-Dispatch-Simulator/repo/##'
