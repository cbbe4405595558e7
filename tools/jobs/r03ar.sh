#!/bin/bash
cd $GRAFT_REPO_ROOT
for n1 in 512 486 540 567 588 630 756 441 420; do
echo "split<=$n1"; BBT_GEN_SPLIT_N1=$n1 timeout -k 10 200 python3 tools/bench_generic.py 2>/dev/null | grep "MHz" | cut -c1-75
done
