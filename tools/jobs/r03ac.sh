#!/bin/bash
# four-step twiddles in the column passes: parity, then same-box A/B of the headline and configs 2, 5
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ac
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
run () { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --no-cpu --no-host-path > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python3 -c "
import json
d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['frac'], d['verified'])"; }
for r in 1 2 3; do
run col_$r BBT_OSM_TW_COL=1
run row_$r BBT_OSM_TW_COL=0
done
for c in config2 config5; do
for v in 1 0; do
BBT_OSM_TW_COL=$v timeout -k 10 200 python3 tools/bench_one.py $c 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/twcol=$v /"
done; done
