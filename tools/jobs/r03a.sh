#!/bin/bash
# round 3, experiment a: stage schedule vs lanes, event overhead, write-through work stores, timeline
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config2 or fused or many_blocks or piecewise or config5 or one_stream" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -3 $OUT/pytest.log
run () {   # name, env..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout -k 10 300 python3 bench.py --no-cpu --no-verify --steps 20 "$@" > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'],d['roofline'].get('pass_ms_per_block'))" | tee -a $OUT/summary.txt
}
WT=$GRAFT_REPO_ROOT/build/libbbt_wt2.so
for r in 1 2; do
run lanes_$r BBT_OSM_SCHED=lanes --
run lanes_noev_$r BBT_OSM_SCHED=lanes -- --no-kernel-timing
run stages_$r X=1 --
run stages_noev_$r X=1 -- --no-kernel-timing
run stages_4x3_$r BBT_OSM_LANES=4 BBT_OSM_CHUNK=3 --
run stages_3x3_$r BBT_OSM_LANES=3 BBT_OSM_CHUNK=3 --
run stages_3x5_$r BBT_OSM_LANES=3 BBT_OSM_CHUNK=5 --
run stages_6x2_$r BBT_OSM_LANES=6 BBT_OSM_CHUNK=2 --
run stages_4x4_$r BBT_OSM_LANES=4 BBT_OSM_CHUNK=4 --
run lanes_wt_$r BBT_OSM_SCHED=lanes BBT_HIP_LIB=$WT --
run stages_wt_$r BBT_HIP_LIB=$WT --
done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_stages -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-verify > $R/$OUT/prof_stages.log 2>&1
echo "prof rc=$?"
cd $R
python3 tools/timeline.py $OUT/prof_stages/run_results.db $OUT/timeline_stages.json > $OUT/timeline_stages.txt 2>&1
tail -30 $OUT/timeline_stages.txt
