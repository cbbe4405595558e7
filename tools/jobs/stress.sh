#!/bin/bash
# randomised geometry tests with more seeds (BBT_TEST_SEED)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stress
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "larger_than_the_padding" > gpurun_out/stress/padding.txt 2>&1; tail -3 gpurun_out/stress/padding.txt
for s in ${SEEDS:-0 1 2 3 4 5 6 7 8}; do
  BBT_TEST_SEED=$s timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_" > gpurun_out/stress/seed$s.txt 2>&1
  tail -1 gpurun_out/stress/seed$s.txt
done
