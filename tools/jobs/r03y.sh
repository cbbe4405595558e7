#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03y
mkdir -p $OUT
timeout -k 10 300 python3 tools/pcie_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie.txt
echo "== no sdma"
HSA_ENABLE_SDMA=0 timeout -k 10 300 python3 tools/pcie_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie_nosdma.txt
for run in 4 8 16 32; do
timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run $run 2>/dev/null | grep pipelined | tee -a $OUT/host_path.jsonl
done
echo "== no sdma"
HSA_ENABLE_SDMA=0 timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 16 2>/dev/null | grep pipelined | tee -a $OUT/host_path_nosdma.jsonl
