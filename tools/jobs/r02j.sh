#!/bin/bash
# GPU job r02j: final profiles of every config + host-sanitizer run of GPU tests
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ASAN=$(bash tools/build_sanitize.sh --runtime)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 BBT_HIP_LIB=$PWD/build/libbbt_hip_asan.so \
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
  -k "small_ or piecewise or random_ or few_channels or odd_and_multi or short_channelizer or not_powers_of_two or pool or rccl or vdif or dada" > $OUT/pytest_asan.log 2>&1; echo "asan rc=$?" | tee -a $OUT/status.txt
tail -4 $OUT/pytest_asan.log
bash tools/jobs/profile_all.sh r02j
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/status.txt
BBT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --blocks 192 --no-cpu > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?" | tee -a $OUT/status.txt
BBT_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --workload config4 --subbands-per-rank 2 --steps 2 --warmup 1 --blocks 2 > $OUT/bench_c4_gloo2.json 2> $OUT/bench_c4_gloo2.err; echo "c4gloo2 rc=$?" | tee -a $OUT/status.txt
timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --blocks 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?" | tee -a $OUT/status.txt
for f in bench bench_gloo2 bench_c4_gloo2 bench_c4; do python3 -c "
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f', d['n_gpus'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_path']['frac'], d['verified'], d.get('with_gather'), (d.get('cpu_baseline') or {}).get('value'))"; done
