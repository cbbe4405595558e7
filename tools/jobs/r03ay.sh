#!/bin/bash
cd $GRAFT_REPO_ROOT
run () { name=$1; shift; env "$@" timeout -k 10 200 python3 tools/bench_next.py f4_ipfb --reps 10 2>/dev/null | grep -o '"row": "[a-z0-9_]*", "munits_per_s": [0-9.]*' | sed "s/^/$name /"; }
for r in 1 2; do
run unfused BBT_FUSE_DECHANNELIZE=0
run fused_v1 BBT_FUSE_DECHANNELIZE=1 BBT_IPFB_VARIANT=1
run fused_v2 BBT_FUSE_DECHANNELIZE=1 BBT_IPFB_VARIANT=2
run fused_v2_l3 BBT_FUSE_DECHANNELIZE=1 BBT_IPFB_VARIANT=2 BBT_IPFB_LANES=3
done
