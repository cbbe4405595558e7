#!/bin/bash
cd $GRAFT_REPO_ROOT
run () { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --no-cpu --no-host-path --no-traffic --no-kernel-timing --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['verified']['ok'])"; }
for r in 1 2 3; do
run 2x6 BBT_X=0
run 2x7 BBT_OSM_CHUNK=7
run 3x4 BBT_OSM_LANES=3 BBT_OSM_CHUNK=4
run 3x5 BBT_OSM_LANES=3 BBT_OSM_CHUNK=5
done
for c in config2; do
BBT_X=0 timeout -k 10 200 python3 tools/bench_one.py $c 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/2x6 /"
BBT_OSM_CHUNK=7 timeout -k 10 200 python3 tools/bench_one.py $c 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/2x7 /"
BBT_OSM_LANES=3 BBT_OSM_CHUNK=4 timeout -k 10 200 python3 tools/bench_one.py $c 2>/dev/null | grep -o '"config": "[a-z0-9]*", "msamples_per_s": [0-9.]*' | sed "s/^/3x4 /"
done
