#!/bin/bash
# inverse filter bank with the dechannelizer staged block by block: parity tests, then the bench row per variant
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03aa
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pfb_convolution_gpu.py -m gpu -q -x -k "inverse or ipfb or Inverse or pfb" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.log
run () { name=$1; shift; env "$@" timeout -k 10 200 python3 tools/bench_next.py f4_ipfb f4_dechan --reps 10 > $OUT/$name.jsonl 2> $OUT/$name.err; echo "$name rc=$?"; grep -o '"row": "[a-z0-9_]*", "munits_per_s": [0-9.]*' $OUT/$name.jsonl; }
for r in 1 2; do
run unfused_$r BBT_FUSE_DECHANNELIZE=0
run v0_$r BBT_IPFB_VARIANT=0
run v1_$r BBT_IPFB_VARIANT=1
run v2_$r BBT_IPFB_VARIANT=2
done
run v0_c2 BBT_IPFB_VARIANT=0 BBT_IPFB_CHUNK=2
run v0_l1 BBT_IPFB_VARIANT=0 BBT_IPFB_LANES=1
run v0_l3 BBT_IPFB_VARIANT=0 BBT_IPFB_LANES=3
run v2_l3 BBT_IPFB_VARIANT=2 BBT_IPFB_LANES=3
