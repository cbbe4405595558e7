#!/bin/bash
# randomised geometry tests with more seeds (BBT_TEST_SEED)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stress
for s in 1 2 3 4 5 6 7 8; do
  BBT_TEST_SEED=$s timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_" 2>&1 | tail -2 | tee -a gpurun_out/stress/log.txt
done
