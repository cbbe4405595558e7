#!/bin/bash
# round 3, experiment c: fused last+first column kernel (CA) with staggered lanes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config2 or fused or many_blocks or piecewise or random" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -3 $OUT/pytest.log
run () {   # name, env..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout -k 10 300 python3 bench.py --no-cpu --steps 20 "$@" > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'],d['roofline'].get('pass_ms_per_block'), (d.get('verified') or {}).get('rel_l2'))" | tee -a $OUT/summary.txt
}
for r in 1 2; do
run base_$r BBT_OSM_CA=0 --
run ca_$r X=1 --
run ca_nostagger_$r BBT_OSM_STAGGER=0 --
run ca_2x5_$r BBT_OSM_CHUNK=5 --
run ca_2x4_$r BBT_OSM_CHUNK=4 --
run ca_3x4_$r BBT_OSM_LANES=3 BBT_OSM_CHUNK=4 --
run ca_2x7_$r BBT_OSM_CHUNK=7 --
run ca_2x8_$r BBT_OSM_CHUNK=8 --
run ca_noev_$r X=1 -- --no-kernel-timing
done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_ca -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-verify --no-kernel-timing > $R/$OUT/prof_ca.log 2>&1
echo "prof rc=$?"
cd $R
python3 tools/timeline.py $OUT/prof_ca/run_results.db $OUT/timeline_ca.json > $OUT/timeline_ca.txt 2>&1
tail -30 $OUT/timeline_ca.txt
