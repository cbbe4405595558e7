#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03w
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
for v in hip maxr12 maxr10; do
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic_$v.txt 2>&1
echo "== $v"; grep -v "amdgpu.ids\|1000 MHz\|1400 MHz" $OUT/generic_$v.txt
done
