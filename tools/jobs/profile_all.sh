#!/bin/bash
# rocprofv3 profiles of every BASELINE config: kernel stats + FETCH_SIZE / WRITE_SIZE (separate passes).
#   bash tools/jobs/profile_all.sh <tag>      -> gpurun_out/<tag>/prof/<workload>/{stats,fetch,write}/run_results.db
set -o pipefail
TAG=${1:-r02e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
run3 () {   # name, command...
    local name=$1; shift
    mkdir -p $OUT/prof/$name
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/$name/stats -o run -- "$@" > $OUT/prof/$name/stats.log 2>&1; echo "$name stats rc=$?" | tee -a $OUT/status.txt
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof/$name/fetch -o run -- "$@" > $OUT/prof/$name/fetch.log 2>&1; echo "$name fetch rc=$?" | tee -a $OUT/status.txt
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof/$name/write -o run -- "$@" > $OUT/prof/$name/write.log 2>&1; echo "$name write rc=$?" | tee -a $OUT/status.txt
}
run3 headline python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-verify
run3 config4 python3 $R/bench.py --workload config4 --steps 2 --warmup 1 --no-verify
cd $R
for c in config1 config2 config3 config5; do
    cd /tmp
    run3 $c python3 $R/tools/bench_one.py $c --reps 4
done
cd $R
for c in config1 config2 config3 config5; do timeout -k 10 200 python3 tools/bench_one.py $c; done > $OUT/bench_one.jsonl 2>$OUT/bench_one.err
cat $OUT/bench_one.jsonl
ls -la $OUT/prof/*/*/ | grep -c run_results
