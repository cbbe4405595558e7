#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ao
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "streams or subband or config4 or random or geometr" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
