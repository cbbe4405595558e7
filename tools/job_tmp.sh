cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r05a
timeout -k 10 300 python3 -u tools/bench_generic.py 2>&1 | grep -v amdgpu.ids | tee $O/generic_lengths.txt
timeout -k 10 500 python3 -u bench.py --workload config4 --steps 3 --warmup 2 --no-traffic > $O/bench_config4.json 2> $O/bench_config4.err; echo "c4 rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_config4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline'), d['config'])"
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/c4_timeline -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload config4 --steps 3 --warmup 1 --no-cpu --no-verify --no-traffic --no-kernel-timing > $O/c4_timeline.log 2>&1; echo "c4 timeline rc=$?" )
python3 tools/timeline.py $O/c4_timeline/run_results.db $O/c4_timeline.json > $O/c4_timeline.txt 2>&1; tail -25 $O/c4_timeline.txt
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ded800 -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_generic.py ded:800 > $O/ded800.log 2>&1; echo "ded800 rc=$?" )
python3 tools/rocprof_db.py stats $O/ded800/run_results.db $O/ded800_kernel_stats.csv | head -6
