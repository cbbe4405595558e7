#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03o
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
run () {
    local name=$1; shift
    env "$@" timeout -k 10 300 python3 tools/bench_generic.py > $OUT/$name.txt 2>&1
    echo "== $name"; grep -v "amdgpu.ids\|1000 MHz\|1400 MHz" $OUT/$name.txt
}
run ept16_ct16 X=1
run ept8_ct16 BBT_HIP_LIB=$L/libbbt_ept8.so
run ept8_ct4 BBT_HIP_LIB=$L/libbbt_ept8.so BBT_GEN_CT=4
run ept8_ct8 BBT_HIP_LIB=$L/libbbt_ept8.so BBT_GEN_CT=8
run ept8_ct4_n1024 BBT_HIP_LIB=$L/libbbt_ept8.so BBT_GEN_CT=4 BBT_GEN_SPLIT_N1=1024
run ept8_ct8_n256 BBT_HIP_LIB=$L/libbbt_ept8.so BBT_GEN_CT=8 BBT_GEN_SPLIT_N1=256
run ept8_old BBT_HIP_LIB=$L/libbbt_ept8.so BBT_GEN_SMALL_RADICES=1 BBT_GEN_SPLIT_N1=0 BBT_GEN_CT=4
