#!/bin/bash
# GPU job r02d: suite + config 4 (three-level fused / two-level) + config 5 prefilter + headline
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_config4_share_of_one_rank > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -25 $OUT/pytest.log
timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?" | tee -a $OUT/status.txt
BBT_FUSE=0 timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify > $OUT/bench_c4_unfused.json 2> $OUT/bench_c4_unfused.err; echo "c4unf rc=$?" | tee -a $OUT/status.txt
for f in bench_c4 bench_c4_unfused; do python3 -c "
import json,sys
d=json.load(open('$OUT/$f.json'))
print('$f', d['value'], d['ms_per_step'], d['roofline']['pass_ms_per_block_isolated'], d['roofline_path']['frac'], d.get('verified'))
"; done
timeout -k 10 600 python3 tools/bench_configs.py > $OUT/configs.txt 2>&1; echo "configs rc=$?" | tee -a $OUT/status.txt
BBT_FUSE_PREFILTER=0 timeout -k 10 600 python3 tools/bench_configs.py config5 > $OUT/configs_nofuse.txt 2>&1; echo "configs_nofuse rc=$?" | tee -a $OUT/status.txt
cat $OUT/configs.txt; cat $OUT/configs_nofuse.txt
timeout -k 10 400 python3 bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/status.txt
python3 -c "
import json
d=json.load(open('$OUT/bench.json')); print(d['value'], d['verified'])"
