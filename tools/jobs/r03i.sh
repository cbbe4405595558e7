#!/bin/bash
# round 3, experiment i: four-step twiddles from tables in the row pass
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03i
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config2 or fused or many_blocks or piecewise or random or one_stream or short_channelizer or few" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -3 $OUT/pytest.log
run () {
    local name=$1; shift
    env "$@" timeout -k 10 300 python3 bench.py --no-cpu --steps 20 > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'],d['roofline'].get('pass_ms_per_block'),d['roofline'].get('pass_ms_per_block_isolated'), (d.get('verified') or {}).get('rel_l2'))" | tee -a $OUT/summary.txt
}
for r in 1 2 3; do
run old_$r BBT_OSM_TW4_TABLES=0
run tab_$r BBT_OSM_TW4_TABLES=1
done
for c in config2 config5; do
for v in 0 1; do
BBT_OSM_TW4_TABLES=$v timeout -k 10 200 python3 tools/bench_one.py $c > $OUT/${c}_tw$v.json 2>$OUT/${c}_tw$v.err
python3 -c "import json;d=json.load(open('$OUT/${c}_tw$v.json'));print('$c tables=$v',d['msamples_per_s'],d['roofline_frac'])" | tee -a $OUT/summary.txt
done
done
