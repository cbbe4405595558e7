#!/bin/bash
# run one test file on the GPU box: F=<path> [K=<expr>]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/file
timeout -k 10 ${T:-900} python3 -m pytest $F -m gpu -q ${K:+-k "$K"} > gpurun_out/file/log.txt 2>&1
tail -15 gpurun_out/file/log.txt
