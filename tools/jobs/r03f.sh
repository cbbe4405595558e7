#!/bin/bash
# round 3, experiment f: short-block route of Convolve (config 5's resampler)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_sampling_gpu.py tests/test_reference_pfb_convolution_gpu.py tests/test_reference_delay_gpu.py -m gpu -q -x -k "convol or Convol or resampl or Resampl or config5 or shift or Shift or delay or Delay" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -5 $OUT/pytest.log
one () {  # name, config, env...
    local name=$1; shift
    local cfg=$1; shift
    env "$@" timeout -k 10 200 python3 tools/bench_one.py $cfg > $OUT/$name.json 2>$OUT/$name.err
    python3 -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['msamples_per_s'],d['roofline_frac'])" | tee -a $OUT/summary.txt
}
for r in 1 2; do
one c5_fir_$r config5 BBT_SHORT_BLOCK=0
one c5_s1024_$r config5 BBT_SHORT_BLOCK=1024
one c5_s2048_$r config5 BBT_SHORT_BLOCK=2048
one c5_s4096_$r config5 BBT_SHORT_BLOCK=4096
done
timeout -k 10 300 python3 tools/bench_configs.py config5 > $OUT/configs5_short.txt 2>&1
BBT_SHORT_BLOCK=0 timeout -k 10 300 python3 tools/bench_configs.py config5 > $OUT/configs5_fir.txt 2>&1
cat $OUT/configs5_short.txt $OUT/configs5_fir.txt
