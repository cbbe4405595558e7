#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/next
timeout -k 10 900 python3 tools/bench_next.py $ROWS > gpurun_out/next/rows.jsonl 2> gpurun_out/next/err.txt
cat gpurun_out/next/rows.jsonl; tail -5 gpurun_out/next/err.txt
