#!/bin/bash
# A/B of environment settings on the headline bench: ENV_A / ENV_B are "VAR=value ..." strings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/envab
for r in 1 2 3; do
  for v in A B; do
    name=ENV_$v
    env ${!name} timeout -k 10 300 python3 bench.py --no-cpu --no-verify --steps 20 ${ARGS} > gpurun_out/envab/$v$r.json 2>gpurun_out/envab/$v$r.err || exit 1
    python3 -c "import json;d=json.load(open('gpurun_out/envab/$v$r.json'));print('$v',d['value'],d['roofline'].get('pass_ms_per_block'))"
  done
done
