#!/bin/bash
# SQ counters of the dominant kernels: headline row/column passes, PFB, FIR (two passes of 8 counters each)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r02k}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
run2 () { local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc $C1 -d $OUT/$name/sq1 -o run -- "$@" > $OUT/$name.sq1.log 2>&1; echo "$name sq1 rc=$?" | tee -a $OUT/status.txt
  timeout -k 10 300 rocprofv3 --pmc $C2 -d $OUT/$name/sq2 -o run -- "$@" > $OUT/$name.sq2.log 2>&1; echo "$name sq2 rc=$?" | tee -a $OUT/status.txt
}
run2 headline python3 $R/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify
[ -n "$ONLY_HEADLINE" ] || run2 config3 python3 $R/tools/bench_one.py config3 --reps 2 --blocks 96
[ -n "$ONLY_HEADLINE" ] || run2 config5 python3 $R/tools/bench_one.py config5 --reps 2 --blocks 12
[ -n "$ONLY_HEADLINE" ] || run2 config2 python3 $R/tools/bench_one.py config2 --reps 2 --blocks 96
