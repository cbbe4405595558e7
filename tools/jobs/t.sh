#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "real_streams_on_block" 2>&1 | tail -25
