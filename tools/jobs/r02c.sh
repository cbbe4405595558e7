#!/bin/bash
# GPU job r02c: 4096 x N2 path (config 4), few-channel fused channelizer, generic path fixes
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "few_channels or longer_than_2_20 or config4" > $OUT/pytest_new.log 2>&1; echo "new rc=$?" | tee -a $OUT/status.txt
tail -25 $OUT/pytest_new.log
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -25 $OUT/pytest.log
timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?" | tee -a $OUT/status.txt
BBT_COL4096_PP=1 timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify > $OUT/bench_c4_pp1.json 2> $OUT/bench_c4_pp1.err; echo "c4pp1 rc=$?" | tee -a $OUT/status.txt
BBT_COL4096_PP=2 timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify > $OUT/bench_c4_pp2.json 2> $OUT/bench_c4_pp2.err; echo "c4pp2 rc=$?" | tee -a $OUT/status.txt
BBT_OSM_THREE_LEVEL=1 BBT_FUSE=0 timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify > $OUT/bench_c4_old.json 2> $OUT/bench_c4_old.err; echo "c4old rc=$?" | tee -a $OUT/status.txt
for f in bench_c4 bench_c4_pp1 bench_c4_pp2 bench_c4_old; do python3 -c "
import json,sys
d=json.load(open('$OUT/$f.json'))
print('$f', d['value'], d['ms_per_step'], d['roofline']['pass_ms_per_block'], d['roofline_path']['frac'], d.get('verified'))
"; done
