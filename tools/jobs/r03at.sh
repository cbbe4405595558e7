#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03at
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_dispersion_gpu.py -m gpu -q -x -k "not_powers_of_two or default or d8 or generic or dispers or geometr" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
timeout -k 10 300 python3 tools/bench_generic.py 2>&1 | grep "MHz" | cut -c1-80
timeout -k 10 300 python3 tools/bench_generic.py 2>&1 | grep "MHz" | cut -c1-80
