#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ab
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in 0 1 2; do
BBT_IPFB_VARIANT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats$v -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 4 > $OUT/stats$v.log 2>&1; echo "stats rc=$?"
python3 $R/tools/rocprof_db.py stats $OUT/stats$v/run_results.db $OUT/kernel_stats$v.csv | head -6
done
BBT_IPFB_VARIANT=1 BBT_IPFB_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats1l1 -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 4 > $OUT/stats1l1.log 2>&1; echo "stats rc=$?"
python3 $R/tools/rocprof_db.py stats $OUT/stats1l1/run_results.db $OUT/kernel_stats1l1.csv | head -6
