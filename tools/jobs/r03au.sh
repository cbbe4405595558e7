#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03au
mkdir -p $OUT
BBT_HIP_LIB=$L/libbbt_pow.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pfb_convolution_gpu.py tests/test_reference_sampling_gpu.py -m gpu -q -x -k "inverse or ipfb or Inverse or convol or Convol or resample or Resample or config5 or short or random_overlap" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $OUT/pytest.log
for r in 1 2; do
for v in hip pow; do
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 200 python3 tools/bench_next.py f4_ipfb --reps 10 2>/dev/null | grep -o '"row": "[a-z0-9_]*", "munits_per_s": [0-9.]*' | sed "s/^/$v /"
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 200 python3 tools/bench_one.py config5 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/$v /"
done; done
