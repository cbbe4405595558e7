#!/bin/bash
# After `gpurun -- 'TAG=<tag> bash tools/jobs/final.sh'`: turn gpurun_out/<tag>/ into the tracked
# summaries under profiles/ (run here, in the repo root).     bash tools/jobs/summarise.sh <tag>
set -e
TAG=${1:?tag}
G=gpurun_out/$TAG
for w in headline config1 config2 config3 config4 config5; do
  python3 tools/rocprof_db.py stats $G/prof/$w/stats/run_results.db profiles/${TAG}_${w}_kernel_stats.csv > /dev/null
  python3 tools/rocprof_db.py traffic $G/prof/$w/fetch/run_results.db $G/prof/$w/write/run_results.db $w $TAG profiles/${TAG}_${w}_traffic.json > /dev/null
done
cp $G/bench.json profiles/${TAG}_bench.json
cp $G/bench_c4.json profiles/${TAG}_bench_config4.json
cp $G/bench_gloo2.json profiles/${TAG}_bench_rehearsal_gloo2.json
cp $G/bench_one.jsonl profiles/${TAG}_bench_one.jsonl
[ -f $G/next_rows.jsonl ] && cp $G/next_rows.jsonl profiles/${TAG}_next_rows.jsonl
ls profiles | grep ${TAG}_ | wc -l
