#!/bin/bash
# GPU job r02b: generic (non power-of-two) path tests + whole suite
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not_powers_of_two or default_geometry or default_block or engine_seam" > $OUT/pytest_gen.log 2>&1; echo "gen rc=$?" | tee -a $OUT/status.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -15 $OUT/pytest_gen.log
tail -15 $OUT/pytest.log
