#!/bin/bash
# round-5 evidence, part A: whole GPU suite, smoke, default bench line, per-config rows
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 700 python3 -u -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_tail.txt
timeout -k 10 200 python3 -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout -k 10 600 python3 -u bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_path'), d['cpu_baseline'], d.get('verified'))"
for c in config1 config2 config3 config5; do timeout -k 10 200 python3 tools/bench_one.py $c; done > $O/bench_one.jsonl 2> $O/bench_one.err; cat $O/bench_one.jsonl
timeout -k 10 500 python3 -u tools/bench_next.py > $O/next_rows.jsonl 2> $O/next_rows.err; echo "next rc=$?"; cat $O/next_rows.jsonl | cut -c1-200
