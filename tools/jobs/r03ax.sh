#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ax
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config4 or subband or 2_24 or long or level" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $OUT/pytest.log
for r in 1 2; do
for v in prev hip; do
BBT_HIP_LIB=$L/libbbt_$v.so timeout -k 10 300 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-kernel-timing --no-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v config4', d['value'], d['verified']['ok'], d['verified']['rel_l2'])"
done; done
