#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03at
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic.txt 2>&1; grep -v amdgpu.ids $OUT/generic.txt
