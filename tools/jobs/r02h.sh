#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in 0 1; do
BBT_OSM_BIG_LANES=$v timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --blocks 4 > $OUT/bench_c4_l$v.json 2> $OUT/bench_c4_l$v.err; echo "c4 lanes$v rc=$?" | tee -a $OUT/status.txt
python3 -c "
import json
d=json.load(open('$OUT/bench_c4_l$v.json'))
print('BIG_LANES=$v', d['value'], d['ms_per_step'], d['roofline']['pass_ms_per_block'], d['roofline_path']['frac'], d['verified']['ok'])"
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "longer_than or config4_subband" 2>&1 | tail -3
