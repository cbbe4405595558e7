#!/bin/bash
# round-5 evidence, part B: the parity file under each route switch, with the final library
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
for e in ${SWITCHES:-BBT_DEFER=0 BBT_HOST_PIPELINE=0 BBT_DEVICE_CHIRP=0}; do
    echo "== $e" | tee -a $O/alt_switches.txt
    env $e timeout -k 10 500 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3 | tee -a $O/alt_switches.txt
done
