#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/dbg
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -o faulthandler_timeout=100 -k "longer_than_2_20 and True" > $OUT/dbg.log 2>&1
echo rc=$?
tail -60 $OUT/dbg.log
