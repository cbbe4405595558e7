#!/bin/bash
# GPU job r02f: plain row pass at 3 waves/SIMD (spills) vs before; suite
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log
for c in config2 config5 config1 config3; do timeout -k 10 200 python3 tools/bench_one.py $c; done 2>/dev/null | tee $OUT/bench_one.jsonl
timeout -k 10 300 python3 bench.py --no-cpu --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['roofline']['pass_ms_per_block_isolated'])"
