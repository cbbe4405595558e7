#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03s
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
for r in 1 2; do
for v in hip dbg32; do
    BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 300 python3 bench.py --no-cpu --no-verify --no-host-path --steps 20 > $OUT/${v}_$r.json 2>$OUT/${v}_$r.err
    python3 -c "import json;d=json.load(open('$OUT/${v}_$r.json'));print('$v',d['value'],d['roofline'].get('pass_ms_per_block'),d['roofline'].get('pass_ms_per_block_isolated'))" | tee -a $OUT/summary.txt
done
done
for c in config1 config3; do for v in hip dbg32; do
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 200 python3 tools/bench_one.py $c > $OUT/${c}_$v.json 2>$OUT/${c}_$v.err
python3 -c "import json;d=json.load(open('$OUT/${c}_$v.json'));print('$c $v',d['msamples_per_s'],d['roofline_frac'])" | tee -a $OUT/summary.txt
done; done
