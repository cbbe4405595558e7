#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/gen
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
BBT_OSM_TWO_LEVEL=1 timeout -k 10 500 python3 tools/bench_generic.py 2>&1 | grep -v amdgpu.ids | grep "MHz" | tee $OUT/generic_two_level.txt
