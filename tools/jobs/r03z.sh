#!/bin/bash
# where the inverse filter bank's time goes: kernel stats + FETCH/WRITE traffic of tools/bench_next.py f4_ipfb
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 4 > $OUT/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 4 > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 4 > $OUT/write.log 2>&1; echo "write rc=$?"
cd $R
python3 tools/rocprof_db.py stats $OUT/stats/run_results.db $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
python3 tools/rocprof_db.py traffic $OUT/fetch/run_results.db $OUT/write/run_results.db ipfb r03z $OUT/traffic.json > $OUT/traffic.txt 2>&1
git checkout profiles/traffic_latest.json 2>/dev/null
head -12 $OUT/kernel_stats.txt
cat $OUT/traffic.txt | head -40
tail -2 $OUT/stats.log
