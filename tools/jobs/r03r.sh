#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03r
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_dispersion_gpu.py tests/test_reference_channelize_gpu.py -m gpu -q -x -k "not_powers or default or golden or engine or seam or reference or channel_count or fft" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -4 $OUT/pytest.log
for v in 0 1; do
BBT_GEN_TW_LDS=$v timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic_tw$v.txt 2>&1
echo "== tw_lds=$v"; grep -v "amdgpu.ids\|1000 MHz\|1400 MHz" $OUT/generic_tw$v.txt
done
timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic_auto.txt 2>&1
echo "== auto"; grep -v "amdgpu.ids\|1000 MHz\|1400 MHz" $OUT/generic_auto.txt
