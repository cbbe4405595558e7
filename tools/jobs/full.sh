#!/bin/bash
# whole GPU suite + default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > gpurun_out/full/pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/full/pytest.txt
[ $rc -eq 0 ] && timeout -k 10 600 python3 bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err && cat gpurun_out/full/bench.json
