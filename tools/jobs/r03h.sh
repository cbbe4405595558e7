#!/bin/bash
# round 3, experiment h: config 4: streaming work-buffer accesses in the outer column passes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03h
mkdir -p $OUT
run () {
    local name=$1; shift
    env "$@" timeout -k 10 300 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify --no-kernel-timing > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'],d['roofline_path']['frac'])" | tee -a $OUT/summary.txt
}
for r in 1 2; do
run base_$r BBT_OSM_MID_MIB=0
run mid128_$r BBT_OSM_MID_MIB=128
run mid128_nt_$r BBT_OSM_MID_MIB=128 BBT_OSM_WORK_NT=1
run mid64_nt_$r BBT_OSM_MID_MIB=64 BBT_OSM_WORK_NT=1
run mid256_nt_$r BBT_OSM_MID_MIB=256 BBT_OSM_WORK_NT=1
run mid128_nt_1lane_$r BBT_OSM_MID_MIB=128 BBT_OSM_WORK_NT=1 BBT_OSM_LANES=1
run mid256_nt_1lane_$r BBT_OSM_MID_MIB=256 BBT_OSM_WORK_NT=1 BBT_OSM_LANES=1
done
