#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_c21.txt; : > $O
python profiles/ab.py $PWD/build/libvds_prev.so $PWD/build/libvds_w7.so $PWD/build/libvds_cat.so --days 400 --rounds 3 >> $O 2>&1
python profiles/ab.py $PWD/build/libvds_prev.so $PWD/build/libvds_w7.so $PWD/build/libvds_cat.so --days 200 --rounds 2 --distinct 16 >> $O 2>&1
VDS_LIB=$PWD/build/libvds_cat.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 >> $O
grep -v amdgpu.ids $O
