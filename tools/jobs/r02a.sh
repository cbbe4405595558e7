#!/bin/bash
# GPU job r02a: suite + new bench modes + SQ counters of the row pass (baseline of round 2).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 - > $OUT/host.txt 2>&1 <<'PY'
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/memory.max'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a', e)
print(open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0])
PY
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/status.txt
BBT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --blocks 192 --no-cpu > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?" | tee -a $OUT/status.txt
timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?" | tee -a $OUT/status.txt
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu --no-verify > $OUT/prof_stats.log 2>&1; echo "stats rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/prof/pmc_sq -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify > $OUT/prof_sq.log 2>&1; echo "sq rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d $OUT/prof/pmc_sq2 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify > $OUT/prof_sq2.log 2>&1; echo "sq2 rc=$?" | tee -a $OUT/status.txt
cat $OUT/status.txt
