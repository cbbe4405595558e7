#!/bin/bash
# A/B of two builds of the library on the headline bench: $A and $B are library paths
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in A B; do
    lib=${!v}
    BBT_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu --no-verify --steps 20 > gpurun_out/ab/$v$r.json 2>gpurun_out/ab/$v$r.err || exit 1
    python3 -c "import json;d=json.load(open('gpurun_out/ab/$v$r.json'));print('$v',d['value'],d['roofline']['pass_ms_per_block'])"
  done
done
