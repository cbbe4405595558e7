#!/bin/bash
# After `gpurun -- 'bash tools/gpu_job.sh evidence TAG=<tag>'`: turn gpurun_out/<tag>/ into the tracked
# summaries under profiles/ (run here, in the repo root).     bash tools/summarise.sh <tag>
set -e
TAG=${1:?tag}
G=gpurun_out/$TAG
for w in headline config1 config2 config3 config4 config5; do
  python3 tools/rocprof_db.py stats $G/prof/$w/stats/run_results.db profiles/${TAG}_${w}_kernel_stats.csv > /dev/null
  python3 tools/rocprof_db.py traffic $G/prof/$w/fetch/run_results.db $G/prof/$w/write/run_results.db $w $TAG profiles/${TAG}_${w}_traffic.json > /dev/null
done
python3 tools/rocprof_db.py pmc $G/prof/headline/sq1/run_results.db $G/prof/headline/sq2/run_results.db profiles/${TAG}_headline_sq_counters.json > /dev/null
cp $G/timeline.txt profiles/${TAG}_headline_timeline.txt
cp $G/timeline.json profiles/${TAG}_headline_timeline.json
cp $G/bench.json profiles/${TAG}_bench.json
cp $G/bench_c4.json profiles/${TAG}_bench_config4.json
cp $G/bench_gloo2.json profiles/${TAG}_bench_rehearsal_gloo2.json
cp $G/bench_one.jsonl profiles/${TAG}_bench_one.jsonl
[ -f $G/next_rows.jsonl ] && cp $G/next_rows.jsonl profiles/${TAG}_next_rows.jsonl
[ -f $G/generic.txt ] && grep -v amdgpu.ids $G/generic.txt > profiles/${TAG}_generic_lengths.txt
[ -f $G/host_path.jsonl ] && cp $G/host_path.jsonl profiles/${TAG}_host_path.jsonl
[ -f $G/pcie.txt ] && grep -v amdgpu.ids $G/pcie.txt > profiles/${TAG}_pcie_probe.txt
[ -f $G/import_order.txt ] && grep -v amdgpu.ids $G/import_order.txt > profiles/${TAG}_import_order.txt
ls profiles | grep ${TAG}_ | wc -l
