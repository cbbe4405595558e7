#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03av
mkdir -p $OUT
head_ () { BBT_HIP_LIB=$L/libbbt_$1.so timeout -k 10 200 python3 bench.py --no-cpu --no-host-path --no-traffic --no-kernel-timing --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 headline', d['value'], d['verified']['ok'], d['verified']['rel_l2'], d['verified']['max_over_rms'])"; }
one_ () { BBT_HIP_LIB=$L/libbbt_$1.so timeout -k 10 200 python3 tools/bench_one.py $2 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/$1 /"; }
ip_ () { BBT_HIP_LIB=$L/libbbt_$1.so timeout -k 10 200 python3 tools/bench_next.py f4_ipfb --reps 10 2>/dev/null | grep -o '"row": "[a-z0-9_]*", "munits_per_s": [0-9.]*' | sed "s/^/$1 /"; }
for r in 1 2 3; do
head_ hip; head_ vrp1; head_ vrp2
done
for r in 1 2; do
one_ hip config3; one_ vpfb1 config3; one_ vpfb2 config3
ip_ hip; ip_ vsm2
one_ hip config5; one_ vsm2 config5
done
BBT_HIP_LIB=$L/libbbt_vpfb1.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pfb_convolution_gpu.py -m gpu -q -k "pfb or Pfb or polyphase or config3" 2>&1 | tail -2
