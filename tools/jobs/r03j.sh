#!/bin/bash
# round 3, experiment j: full suite (pool, unpack, short blocks, host pipeline), host path rate
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03j
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "host_path or device_resident or piecewise or vdif or dada or unpack or inverse_poly" > $OUT/pytest_first.log 2>&1
echo "first rc=$?" | tee $OUT/status.txt
tail -5 $OUT/pytest_first.log
timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 16 > $OUT/host_path.jsonl 2>$OUT/host_path.err
cat $OUT/host_path.jsonl; tail -3 $OUT/host_path.err
timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 8 > $OUT/host_path_r8.jsonl 2>$OUT/host_path_r8.err
cat $OUT/host_path_r8.jsonl
timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 32 > $OUT/host_path_r32.jsonl 2>$OUT/host_path_r32.err
cat $OUT/host_path_r32.jsonl
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -4 $OUT/pytest.log
