#!/bin/bash
# bench.py with roofline.traffic measured live (rocprofv3 children), and the same under rocprofv3 (must skip it)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ad
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt; echo "bench rc=$?"
cat $OUT/time.txt | tail -3
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_note'][:160]); print(d['host_path']); print(d['cpu_baseline'])"
cd /tmp
env | grep -i rocp > $OUT/env_plain.txt
( time timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run -- python3 $R/bench.py --no-cpu --no-host-path --steps 4 > $OUT/bench_prof.json 2> $OUT/bench_prof.err ) 2> $OUT/time_prof.txt; echo "prof rc=$?"
tail -3 $OUT/time_prof.txt
python3 -c "
import json
d=json.loads(open('$OUT/bench_prof.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['traffic'], r['traffic_note'][-120:])"
