#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "convolve or resample or config5 or tones" 2>&1 | tail -3
timeout -k 10 300 python3 tools/bench_configs.py config5 2>/dev/null | tee $OUT/config5.txt
timeout -k 10 200 python3 tools/bench_one.py config5 2>/dev/null | tee $OUT/bench_one.jsonl
timeout -k 10 200 python3 tools/bench_one.py config2 2>/dev/null | tee -a $OUT/bench_one.jsonl
