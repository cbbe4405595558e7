#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -4 $OUT/pytest.log
timeout -k 10 400 python3 tools/cpu_baseline_all.py > $OUT/cpu_baseline_all.log 2>&1; echo "cpu rc=$?" | tee -a $OUT/status.txt
tail -1 $OUT/cpu_baseline_all.log > $OUT/cpu_baseline_all.json
cat $OUT/cpu_baseline_all.log | grep -v "^{" 
