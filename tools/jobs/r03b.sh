#!/bin/bash
# round 3, experiment b: cache policy of the work-buffer stores (lanes schedule)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b
mkdir -p $OUT
run () {   # name, env..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout -k 10 300 python3 bench.py --no-cpu --no-verify --steps 20 "$@" > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'],d['roofline'].get('pass_ms_per_block'))" | tee -a $OUT/summary.txt
}
L=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib
for r in 1 2; do
run plain_$r BBT_OSM_SCHED=lanes --
run nt_$r BBT_OSM_SCHED=lanes BBT_HIP_LIB=$L/libbbt_wt1.so --
run sc1_$r BBT_OSM_SCHED=lanes BBT_HIP_LIB=$L/libbbt_wt2.so --
run sc1nt_$r BBT_OSM_SCHED=lanes BBT_HIP_LIB=$L/libbbt_wt3.so --
run sc1_2x4_$r BBT_OSM_SCHED=lanes BBT_OSM_CHUNK=4 BBT_HIP_LIB=$L/libbbt_wt2.so --
run sc1_3x4_$r BBT_OSM_SCHED=lanes BBT_OSM_LANES=3 BBT_OSM_CHUNK=4 BBT_HIP_LIB=$L/libbbt_wt2.so --
run plain_3x4_$r BBT_OSM_SCHED=lanes BBT_OSM_LANES=3 BBT_OSM_CHUNK=4 --
run sc1_2x3_$r BBT_OSM_SCHED=lanes BBT_OSM_CHUNK=3 BBT_HIP_LIB=$L/libbbt_wt2.so --
run sc1_4x3_$r BBT_OSM_SCHED=lanes BBT_OSM_LANES=4 BBT_OSM_CHUNK=3 BBT_HIP_LIB=$L/libbbt_wt2.so --
done
