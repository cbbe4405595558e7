#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ai
mkdir -p $OUT
for r in 1 2; do
for w in 0 16 32 48; do
BBT_COL_WIDE=$w timeout -k 10 200 python3 tools/bench_one.py config5 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/wide=$w /"
done; done
BBT_COL_WIDE=48 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config5 or streams or subband" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $OUT/pytest.log
