#!/bin/bash
# round 3, experiment e: full GPU suite on the new defaults (sc1 work stores), config 5 lanes/chunk sweep
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03e
mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -3 $OUT/pytest.log
one () {  # name, config, env...
    local name=$1; shift
    local cfg=$1; shift
    env "$@" timeout -k 10 200 python3 tools/bench_one.py $cfg > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['msamples_per_s'],d['roofline_frac'])" | tee -a $OUT/summary.txt
}
for r in 1 2; do
one c5_base_$r config5 X=1
one c5_3x1_$r config5 BBT_OSM_LANES=3 BBT_OSM_CHUNK=1
one c5_1x3_$r config5 BBT_OSM_LANES=1 BBT_OSM_CHUNK=3
one c5_2x2_$r config5 BBT_OSM_LANES=2 BBT_OSM_CHUNK=2
one c5_4x1_$r config5 BBT_OSM_LANES=4 BBT_OSM_CHUNK=1
one c5_1x2_$r config5 BBT_OSM_LANES=1 BBT_OSM_CHUNK=2
one c5_3x2_$r config5 BBT_OSM_LANES=3 BBT_OSM_CHUNK=2
done
one c1 config1 X=1
one c2 config2 X=1
one c3 config3 X=1
timeout -k 10 300 python3 bench.py --steps 20 > $OUT/bench.json 2>$OUT/bench.err
python3 -c "import json;d=json.load(open('$OUT/bench.json'));print('headline',d['value'],d['roofline'],d['verified'])"
