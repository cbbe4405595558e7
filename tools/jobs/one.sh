#!/bin/bash
# run the GPU tests selected by $K (pytest -k expression)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/one
timeout -k 10 ${T:-600} python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" > gpurun_out/one/log.txt 2>&1
tail -5 gpurun_out/one/log.txt
