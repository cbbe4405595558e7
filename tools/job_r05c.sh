#!/bin/bash
# round-5 evidence, part C: rocprofv3 summaries of the headline command (kernel stats, traffic, timeline) and config 4 traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_job.sh prof3 TAG=r05 NAME=headline CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu --no-verify --no-host-path --no-traffic" 2>&1 | tail -3
bash tools/gpu_job.sh timeline TAG=r05 2>&1 | tail -12
bash tools/gpu_job.sh prof3 TAG=r05 NAME=config4 CMD="python3 $GRAFT_REPO_ROOT/bench.py --workload config4 --steps 2 --warmup 2 --no-verify --no-traffic --no-cpu" 2>&1 | tail -3
BBT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --blocks 192 --no-cpu --no-host-path > gpurun_out/r05/bench_gloo2.json 2> gpurun_out/r05/bench_gloo2.err; echo "gloo2 rc=$?"
