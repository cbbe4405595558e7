#!/bin/bash
# round 3 evidence job: suite, profiles of every config (kernel stats, FETCH/WRITE traffic), SQ counters and
# timeline of the headline, bench lines, neighbouring rows, generic engine, host path, PCIe probe
set -o pipefail
TAG=${TAG:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -3 $OUT/pytest.log
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run3 () {   # name, command...
    local name=$1; shift
    mkdir -p $OUT/prof/$name
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/$name/stats -o run -- "$@" > $OUT/prof/$name/stats.log 2>&1; echo "$name stats rc=$?" | tee -a $OUT/status.txt
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof/$name/fetch -o run -- "$@" > $OUT/prof/$name/fetch.log 2>&1; echo "$name fetch rc=$?" | tee -a $OUT/status.txt
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof/$name/write -o run -- "$@" > $OUT/prof/$name/write.log 2>&1; echo "$name write rc=$?" | tee -a $OUT/status.txt
}
run3 headline python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-verify --no-host-path
run3 config4 python3 $R/bench.py --workload config4 --steps 2 --warmup 1 --no-verify
for c in config1 config2 config3 config5; do
    run3 $c python3 $R/tools/bench_one.py $c --reps 4
done
# timeline of the headline without per-kernel events
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/headline/timeline -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-verify --no-host-path --no-kernel-timing > $OUT/prof/headline/timeline.log 2>&1; echo "timeline rc=$?" | tee -a $OUT/status.txt
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
timeout -k 10 300 rocprofv3 --pmc $C1 -d $OUT/prof/headline/sq1 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify --no-host-path > $OUT/prof/headline/sq1.log 2>&1; echo "sq1 rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 rocprofv3 --pmc $C2 -d $OUT/prof/headline/sq2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --blocks 96 --no-cpu --no-verify --no-host-path > $OUT/prof/headline/sq2.log 2>&1; echo "sq2 rc=$?" | tee -a $OUT/status.txt
cd $R
for c in config1 config2 config3 config5; do timeout -k 10 200 python3 tools/bench_one.py $c; done > $OUT/bench_one.jsonl 2>$OUT/bench_one.err
cat $OUT/bench_one.jsonl
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/status.txt
timeout -k 10 500 python3 bench.py --workload config4 --steps 3 --warmup 1 --blocks 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?" | tee -a $OUT/status.txt
timeout -k 10 900 python3 tools/bench_next.py > $OUT/next_rows.jsonl 2> $OUT/next_rows.err; echo "next rc=$?" | tee -a $OUT/status.txt
BBT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --blocks 192 --no-cpu --no-host-path > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 python3 tools/bench_generic.py > $OUT/generic.txt 2>&1; echo "generic rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 python3 tools/bench_host_path.py --blocks 192 --run 16 > $OUT/host_path.jsonl 2>$OUT/host_path.err; echo "host rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 python3 tools/pcie_probe.py > $OUT/pcie.txt 2>&1; echo "pcie rc=$?" | tee -a $OUT/status.txt
for f in bench bench_c4 bench_gloo2; do python3 -c "
import json
d=json.loads(open(\"$OUT/$f.json\").read().strip().splitlines()[-1])
print(\"$f\", d[\"n_gpus\"], d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline_path\"][\"frac\"], d[\"verified\"][\"ok\"], (d.get(\"with_gather\") or {}).get(\"value\"), (d.get(\"cpu_baseline\") or {}).get(\"value\"), (d.get(\"host_path\") or {}).get(\"value\"))"; done
cat $OUT/status.txt
