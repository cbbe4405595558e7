#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03u
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
for r in 1 2; do
for v in hip out1 out2 out3; do
    BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 300 python3 bench.py --no-cpu --no-verify --no-host-path --steps 20 > $OUT/${v}_$r.json 2>$OUT/${v}_$r.err
    python3 -c "import json;d=json.load(open('$OUT/${v}_$r.json'));print('$v',d['value'],d['roofline'].get('pass_ms_per_block'))" | tee -a $OUT/summary.txt
done
done
for v in hip out1 out2 out3; do
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 200 python3 tools/bench_one.py config1 > $OUT/c1_$v.json 2>$OUT/c1_$v.err
python3 -c "import json;d=json.load(open('$OUT/c1_$v.json'));print('config1 $v',d['msamples_per_s'],d['roofline_frac'])" | tee -a $OUT/summary.txt
done
