#!/bin/bash
# round 3, experiment n: generic (non power-of-two) engine: composite radices, tile widths
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03n
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_dispersion_gpu.py tests/test_reference_channelize_gpu.py -m gpu -q -x -k "not_powers or default or golden or engine or seam or reference or channel_count or fft" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/status.txt
tail -4 $OUT/pytest.log
run () {
    local name=$1; shift
    env "$@" timeout -k 10 300 python3 tools/bench_generic.py > $OUT/$name.txt 2>&1
    echo "== $name"; grep -v amdgpu.ids $OUT/$name.txt
}
run old BBT_GEN_SMALL_RADICES=1 BBT_GEN_SPLIT_N1=0
run radix16_balanced BBT_GEN_SPLIT_N1=0
run radix16_n1_512 X=1
run radix12_n1_512 BBT_GEN_MAX_RADIX=12
run radix16_n1_256 BBT_GEN_SPLIT_N1=256
run radix16_n1_1024 BBT_GEN_SPLIT_N1=1024
