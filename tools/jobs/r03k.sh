#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03k
mkdir -p $OUT
timeout -k 10 300 python3 tools/pcie_probe.py > $OUT/pcie.txt 2>&1
cat $OUT/pcie.txt
HSA_ENABLE_SDMA=0 timeout -k 10 300 python3 tools/pcie_probe.py > $OUT/pcie_nosdma.txt 2>&1
cat $OUT/pcie_nosdma.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "host_path or short_blocks or config5 or resampl" > $OUT/pytest_first.log 2>&1
echo "first rc=$?" | tee $OUT/status.txt
tail -5 $OUT/pytest_first.log
timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 16 > $OUT/host_path.jsonl 2>$OUT/host_path.err
cat $OUT/host_path.jsonl
