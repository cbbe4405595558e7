#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03af
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "packed or vdif or dada or host_path" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
timeout -k 10 400 python3 bench.py --no-cpu --no-traffic --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); h=d['host_path']
print(d['value']); print({k:v for k,v in h.items() if k not in ('what','packed_input')}); print(h['packed_input'])"
tail -3 $OUT/bench.err
