#!/bin/bash
# round 3, experiment d: timing-only variants (what is the headline sensitive to?)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
for r in 1 2; do
for v in hip dbg1 dbg2 dbg4 dbg6 dbg8 dbg16; do
    BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 300 python3 bench.py --no-cpu --no-verify --steps 20 > $OUT/${v}_$r.json 2>$OUT/${v}_$r.err
    python3 -c "import json;d=json.load(open('$OUT/${v}_$r.json'));print('$v',d['value'],d['roofline'].get('pass_ms_per_block'),d['roofline'].get('pass_ms_per_block_isolated'))" | tee -a $OUT/summary.txt
done
done
