#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03v
mkdir -p $OUT
run () {
    local name=$1; shift
    env "$@" timeout -k 10 300 python3 bench.py --no-cpu --no-verify --no-host-path --steps 20 > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'])" | tee -a $OUT/summary.txt
}
for r in 1 2; do
run base_$r X=1
run c5_$r BBT_OSM_CHUNK=5
run c7_$r BBT_OSM_CHUNK=7
run c8_$r BBT_OSM_CHUNK=8
run l3c4_$r BBT_OSM_LANES=3 BBT_OSM_CHUNK=4
run l3c5_$r BBT_OSM_LANES=3 BBT_OSM_CHUNK=5
run ca_$r BBT_OSM_CA=1
run ca_c8_$r BBT_OSM_CA=1 BBT_OSM_CHUNK=8
done
for s in 0 256 512 1024 2048; do
BBT_GEN_SPLIT_N1=$s timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic_split$s.txt 2>&1
echo "== split $s"; grep "800 MHz\|600 MHz" $OUT/generic_split$s.txt
done
