#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4ab
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "few_channels or longer_than or random_fused" 2>&1 | tail -3
one () { local tag=$1; shift
  env "$@" timeout -k 10 400 python3 bench.py --workload config4 --steps 3 --warmup 1 --blocks 4 --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['roofline']['pass_ms_per_block_isolated'])" | tee -a $OUT/c4ab.txt
}
one three_waves_dpp BBT_X=0
one two_waves_dpp BBT_HIP_LIB=$PWD/build/libbbt_hip_small2w.so
one unfused BBT_FUSE=0
