#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "plain_c or full_scale_golden or two_ranks" > $OUT/pytest_new.log 2>&1; echo "new rc=$?" | tee -a $OUT/status.txt
tail -15 $OUT/pytest_new.log
