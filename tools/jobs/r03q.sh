#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03q
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/gen800.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import baseband_tasks_amd as bt
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.view_as_complex(torch.randn((48 * 2**20, 2, 2), generator=g, device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=800e6, sideband=1)
dd = bt.Dedisperse(ds, 100.)
dd.max_frames_per_call = 10**6
for _ in range(4):
    dd.invalidate_cache(); dd.seek(0); dd.read_device(dd.shape[0])
torch.cuda.synchronize()
PY
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_gen -o run -- python3 /tmp/gen800.py > $R/$OUT/prof_gen.log 2>&1
cd $R
python3 tools/rocprof_db.py stats $OUT/prof_gen/run_results.db $OUT/gen800_kernel_stats.csv
timeout -k 10 600 python3 tools/bench_next.py > $OUT/next_rows.jsonl 2>$OUT/next_rows.err
cat $OUT/next_rows.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['row'], d['munits_per_s'], d['frac'])"
