#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 0 1 3 9 4 12 5; do
BBT_COL_WIDE=$w timeout -k 10 200 python3 tools/bench_streams.py 16 32 2>/dev/null
done
for w in 0 1; do
BBT_COL_WIDE=$w timeout -k 10 200 python3 tools/bench_streams.py 16 2>/dev/null
done
