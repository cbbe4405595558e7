#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ap
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $R/tools/bench_next.py f1_fused --reps 4 > $OUT/stats.log 2>&1; echo "stats rc=$?"
python3 $R/tools/rocprof_db.py stats $OUT/stats/run_results.db $OUT/kernel_stats.csv | head -8
grep '^{' $OUT/stats.log
