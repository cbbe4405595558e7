#!/bin/bash
# GPU-box jobs, one script:   gpurun -- 'bash tools/gpu_job.sh <job> [TAG=<tag>] [VAR=value ...]'
# Every job writes under gpurun_out/$TAG/ (default tag: the job's name) and joins its GPU steps with
# `&&`-like logic where a failure should stop the rest (status lines go to $OUT/status.txt).
#
#   suite                     whole GPU test suite (-x), then the default bench.py
#   tests   K=<expr> [F=<files>] [T=<s>]     selected tests (pytest -k) of test_gpu_parity.py or files F
#   stress  SEEDS="0 1 .."    the randomised tests under many BBT_TEST_SEEDs + the padding regression + tools/soak_reads.py
#   ab      A=<lib> B=<lib> [ARGS=..] [N=3]  alternating headline runs of two library builds
#   envab   ENV_A=".." ENV_B=".." [ARGS=..] [N=3]   the same for two environment settings
#   one     CONFIGS="config2 config5" [ENVS="A=1|B=2"]   tools/bench_one.py rows (optionally per env)
#   next    ROWS=".."         rows of tools/bench_next.py
#   timeline [ARGS=..]        kernel trace of the headline without timing events -> tools/timeline.py
#   chain   LABEL=<label>     kernel trace of one chain of tools/default_device_reads.py -> tools/chain_timeline.py
#   prof3   NAME=<n> CMD=".." kernel stats + FETCH_SIZE + WRITE_SIZE (three separate runs) of a command
#   sq      [CMD=..]          SQ counters (two passes of 8) of a command (default: headline, 96 blocks)
#   evidence                  round-end: suite, profiles of every config, timeline, SQ counters, bench
#                             lines, next rows, generic lengths, host path (then: tools/summarise.sh)
set -o pipefail
JOB=${1:?job}; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${TAG:-$JOB}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
say () { echo "$@" | tee -a $OUT/status.txt; }
value () { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$2', d.get('value'), 'ms', d.get('ms_per_step'), 'frac', r.get('frac'), 'path', (d.get('roofline_path') or {}).get('frac'), 'pass_ms', r.get('pass_ms_per_block'), 'ok', (d.get('verified') or {}).get('ok'))"; }

prof3 () {   # name, command...   (counters in their own runs, the program directly after --)
    local name=$1; shift
    mkdir -p $OUT/prof/$name
    ( cd /tmp
      timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof/$name/stats -o run -- "$@" > $OUT/prof/$name/stats.log 2>&1; say "$name stats rc=$?"
      timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof/$name/fetch -o run -- "$@" > $OUT/prof/$name/fetch.log 2>&1; say "$name fetch rc=$?"
      timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof/$name/write -o run -- "$@" > $OUT/prof/$name/write.log 2>&1; say "$name write rc=$?" )
}
timeline () {   # extra bench args...
    mkdir -p $OUT/prof/headline
    ( cd /tmp
      timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof/headline/timeline -o run -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu --no-verify --no-host-path --no-traffic --no-kernel-timing "$@" > $OUT/prof/headline/timeline.log 2>&1; say "timeline rc=$?" )
    python3 tools/timeline.py $OUT/prof/headline/timeline/run_results.db $OUT/timeline.json > $OUT/timeline.txt 2>&1
    cat $OUT/timeline.txt
}
sq () {   # name, command...
    local name=$1; shift
    local C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
    local C2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
    mkdir -p $OUT/prof/$name
    ( cd /tmp
      timeout -k 10 400 rocprofv3 --pmc $C1 -d $OUT/prof/$name/sq1 -o run -- "$@" > $OUT/prof/$name/sq1.log 2>&1; say "$name sq1 rc=$?"
      timeout -k 10 400 rocprofv3 --pmc $C2 -d $OUT/prof/$name/sq2 -o run -- "$@" > $OUT/prof/$name/sq2.log 2>&1; say "$name sq2 rc=$?" )
}
suite () {
    timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; local rc=$?
    say "pytest rc=$rc"; tail -3 $OUT/pytest.log
    return $rc
}

case $JOB in
suite)
    suite && timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err && value $OUT/bench.json bench ;;
tests)
    timeout -k 10 ${T:-900} python3 -m pytest ${F:-tests/test_gpu_parity.py} -m gpu -q -x ${K:+-k "$K"} > $OUT/log.txt 2>&1; say "tests rc=$?"
    tail -15 $OUT/log.txt ;;
stress)
    timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "larger_than_the_padding" > $OUT/padding.txt 2>&1; tail -3 $OUT/padding.txt
    for s in ${SEEDS:-0 1 2 3 4 5 6 7 8}; do
        BBT_TEST_SEED=$s timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_" > $OUT/seed$s.txt 2>&1 || { say "seed $s failed"; tail -5 $OUT/seed$s.txt; exit 1; }
        tail -1 $OUT/seed$s.txt
    done
    timeout -k 10 600 python3 tools/soak_reads.py > $OUT/soak.txt 2>&1; say "soak rc=$?"; tail -1 $OUT/soak.txt ;;
ab|envab)
    for r in $(seq 1 ${N:-3}); do
        for v in A B; do
            if [ $JOB = ab ]; then e="BBT_HIP_LIB=${!v}"; else n=ENV_$v; e="${!n}"; fi
            env $e timeout -k 10 300 python3 bench.py --no-cpu --no-verify --no-host-path --no-traffic --steps 20 $ARGS > $OUT/$v$r.json 2> $OUT/$v$r.err || { say "$v$r failed"; tail -5 $OUT/$v$r.err; exit 1; }
            value $OUT/$v$r.json "$v ($e)"
        done
    done ;;
one)
    IFS='|' read -ra EL <<< "${ENVS:-}"
    [ ${#EL[@]} -eq 0 ] && EL=("")
    for r in $(seq 1 ${N:-1}); do
      for e in "${EL[@]}"; do
        for c in ${CONFIGS:-config1 config2 config3 config5}; do
            env $e timeout -k 10 300 python3 tools/bench_one.py $c $ARGS 2>> $OUT/one.err | tee -a $OUT/one.jsonl | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$c [$e]', d.get('msamples_per_s'), 'Msamples/s', d.get('roofline_frac'))" || { say "$c failed"; tail -5 $OUT/one.err; exit 1; }
        done
      done
    done ;;
next)
    timeout -k 10 900 python3 tools/bench_next.py $ROWS > $OUT/rows.jsonl 2> $OUT/err.txt; say "next rc=$?"
    cat $OUT/rows.jsonl; tail -5 $OUT/err.txt ;;
timeline)
    timeline $ARGS ;;
chain)      # LABEL="Dedisperse(Resample": kernel trace of one chain of tools/default_device_reads.py -> tools/chain_timeline.py
    ( cd /tmp
      timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run -- python3 $R/tools/default_device_reads.py "${LABEL:?LABEL}" > $OUT/run.txt 2>&1; say "chain rc=$?" )
    tail -3 $OUT/run.txt
    python3 tools/chain_timeline.py $OUT/prof/run_results.db ${STRETCHES:-2} > $OUT/timeline.txt 2>&1
    cat $OUT/timeline.txt ;;
prof3)
    prof3 ${NAME:?NAME} $CMD ;;
sq)
    sq ${NAME:-headline} ${CMD:-python3 $R/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify --no-host-path --no-traffic} ;;
evidence)
    suite || exit 1
    prof3 headline python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-verify --no-host-path --no-traffic
    prof3 config4 python3 $R/bench.py --workload config4 --steps 2 --warmup 2 --no-verify --no-traffic
    for c in config1 config2 config3 config5; do prof3 $c python3 $R/tools/bench_one.py $c --reps 4; done
    timeline
    sq headline python3 $R/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify --no-host-path --no-traffic
    for c in config1 config2 config3 config5; do timeout -k 10 200 python3 tools/bench_one.py $c; done > $OUT/bench_one.jsonl 2> $OUT/bench_one.err
    cat $OUT/bench_one.jsonl
    timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; say "bench rc=$?"
    timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 2 --blocks 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; say "c4 rc=$?"
    timeout -k 10 900 python3 tools/bench_next.py > $OUT/next_rows.jsonl 2> $OUT/next_rows.err; say "next rc=$?"
    BBT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --blocks 192 --no-cpu --no-host-path > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; say "gloo2 rc=$?"
    timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic.txt 2>&1; say "generic rc=$?"
    timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 16 > $OUT/host_path.jsonl 2> $OUT/host_path.err; say "host rc=$?"
    timeout -k 10 300 python3 tools/pcie_probe.py > $OUT/pcie.txt 2>&1; say "pcie rc=$?"
    timeout -k 10 300 python3 tools/import_order.py torch-first > $OUT/import_order.txt 2>&1 && timeout -k 10 300 python3 tools/import_order.py lib-first >> $OUT/import_order.txt 2>&1; say "import order rc=$?"
    for f in bench bench_c4 bench_gloo2; do value $OUT/$f.json $f; done
    cat $OUT/status.txt ;;
*)
    echo "unknown job $JOB"; exit 2 ;;
esac
