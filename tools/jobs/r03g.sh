#!/bin/bash
# round 3, experiment g: config 4 middle passes in cache-sized pieces; short-block convolve tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_sampling_gpu.py tests/test_reference_pfb_convolution_gpu.py tests/test_reference_delay_gpu.py -m gpu -q -x -k "convol or Convol or resampl or Resampl or config5 or config4 or longer or delay or Delay or subband" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -5 $OUT/pytest.log
for r in 1 2; do
for mib in 0 32 64 128 256; do
    BBT_OSM_MID_MIB=$mib timeout -k 10 300 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify --no-kernel-timing > $OUT/c4_${mib}_$r.json 2>$OUT/c4_${mib}_$r.err
    python3 -c "import json;d=json.load(open('$OUT/c4_${mib}_$r.json'));print('c4 mid_mib=$mib',d['value'],d['roofline_path']['frac'])" | tee -a $OUT/summary.txt
done
done
BBT_OSM_LANES=1 BBT_OSM_MID_MIB=64 timeout -k 10 300 python3 bench.py --workload config4 --steps 3 --warmup 1 --no-verify --no-kernel-timing > $OUT/c4_1lane.json 2>$OUT/c4_1lane.err
python3 -c "import json;d=json.load(open('$OUT/c4_1lane.json'));print('c4 1 lane',d['value'])" | tee -a $OUT/summary.txt
