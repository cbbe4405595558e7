#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03aw
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest.log
