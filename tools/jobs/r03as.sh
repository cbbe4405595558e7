#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03as
mkdir -p $OUT
run () { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --no-cpu --no-host-path --no-traffic --no-kernel-timing --steps 10 > $OUT/$name.json 2> $OUT/$name.err; python3 -c "
import json
d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['verified']['ok'])"; }
for r in 1 2 3; do
run pad0_$r BBT_COL_LDS_PAD_FIRST=0
run pad9k_$r BBT_X=0
done
for c in config2 config5; do
for v in 0 9216 0 9216; do
BBT_COL_LDS_PAD_FIRST=$v timeout -k 10 200 python3 tools/bench_one.py $c 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/pad=$v /"
done; done
