#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_" > $OUT/pytest_random.log 2>&1; echo "random rc=$?" | tee -a $OUT/status.txt
tail -30 $OUT/pytest_random.log
