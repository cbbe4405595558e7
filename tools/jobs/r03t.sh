#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03t
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pfb_convolution_gpu.py -m gpu -q -x -k "pfb or PFB or polyphase or Inversion or config3" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -3 $OUT/pytest.log
for r in 1 2 3; do for v in prev hip; do
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 200 python3 tools/bench_one.py config3 > $OUT/c3_${v}_$r.json 2>$OUT/c3_${v}_$r.err
python3 -c "import json;d=json.load(open('$OUT/c3_${v}_$r.json'));print('config3 $v',d['msamples_per_s'],d['roofline_frac'])" | tee -a $OUT/summary.txt
done; done
for v in prev hip; do
BBT_HIP_LIB=$L/libbbt_$v.so ROWS="pfb_12x256 pfb_16x4096 pfb_4x1024 pfb_8x2048 pfb_real_12x1024" timeout -k 10 300 python3 tools/bench_next.py pfb_12x256 pfb_16x4096 pfb_4x1024 pfb_8x2048 pfb_real_12x1024 > $OUT/next_$v.jsonl 2>$OUT/next_$v.err
python3 -c "
import sys, json
for l in open('$OUT/next_$v.jsonl'):
    d = json.loads(l); print('$v', d['row'], d['munits_per_s'], d['frac'])"
done
