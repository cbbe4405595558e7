#!/bin/bash
# config 4: wide column tiles (8 pairs x 8 / 4 columns) for the outer column passes
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ag
mkdir -p $OUT
run () { name=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-kernel-timing --no-traffic > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python3 -c "
import json
d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['verified']['ok'], d['verified']['rel_l2'])"; }
run base_1 BBT_COL_WIDE=0
run w1_1 BBT_COL_WIDE=1
run w3 BBT_COL_WIDE=3
run w9 BBT_COL_WIDE=9
run w4 BBT_COL_WIDE=4
run w12 BBT_COL_WIDE=12
run base_2 BBT_COL_WIDE=0
run w1_2 BBT_COL_WIDE=1
