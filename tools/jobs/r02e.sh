#!/bin/bash
# GPU job r02e: whole suite, row-pass grid A/B, profiles of every config
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x -v > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log
for v in 1 0; do
  BBT_ROWPASS_REMAP=$v timeout -k 10 300 python3 bench.py --no-cpu --no-verify > $OUT/bench_remap$v.json 2> $OUT/bench_remap$v.err; echo "remap$v rc=$?" | tee -a $OUT/status.txt
  python3 -c "
import json
d=json.load(open('$OUT/bench_remap$v.json')); print('remap$v', d['value'], d['roofline']['pass_ms_per_block'], d['roofline']['pass_ms_per_block_isolated'])"
done
bash tools/jobs/profile_all.sh r02e
