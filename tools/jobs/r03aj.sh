#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03aj
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config5 or planar or host_path or resample or Resample" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest.log
for r in 1; do
for v in 1 0; do
BBT_PLANAR=$v timeout -k 10 200 python3 tools/bench_one.py config5 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/planar=$v /"
done; done
