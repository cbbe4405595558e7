#!/bin/bash
# final job of the round: suite, profiles of every config, bench lines
set -o pipefail
TAG=${TAG:-r02s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -3 $OUT/pytest.log
bash tools/jobs/profile_all.sh $TAG
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/status.txt
timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --blocks 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?" | tee -a $OUT/status.txt
timeout -k 10 900 python3 tools/bench_next.py > $OUT/next_rows.jsonl 2> $OUT/next_rows.err; echo "next rc=$?" | tee -a $OUT/status.txt
BBT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --blocks 192 --no-cpu > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?" | tee -a $OUT/status.txt
for f in bench bench_c4 bench_gloo2; do python3 -c "
import json
d=json.loads(open(\"$OUT/$f.json\").read().strip().splitlines()[-1])
print(\"$f\", d[\"n_gpus\"], d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline_path\"][\"frac\"], d[\"verified\"][\"ok\"], (d.get(\"with_gather\") or {}).get(\"value\"), (d.get(\"cpu_baseline\") or {}).get(\"value\"))"; done
