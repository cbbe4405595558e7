#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ah
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config4 or subband or long or level or many_streams or streams" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
BBT_COL_WIDE=3 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config4 or subband or streams" > $OUT/pytest3.log 2>&1; echo "pytest wide=3 rc=$?"
tail -2 $OUT/pytest3.log
