#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03x
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
x = torch.view_as_complex(torch.randn((24 * 2**20, 2, 2), generator=g, device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=800e6, sideband=1)
dd = bt.Dedisperse(ds, 100.)
dd.max_frames_per_call = 10**6
for _ in range(2):
    dd.invalidate_cache(); dd.seek(0); dd.read_device(dd.shape[0])
torch.cuda.synchronize()
PY
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc $C1 -d $OUT/sq1 -o run -- python3 /tmp/gen800.py > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc $C2 -d $OUT/sq2 -o run -- python3 /tmp/gen800.py > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
python3 tools/rocprof_db.py pmc $OUT/sq1/run_results.db $OUT/sq2/run_results.db $OUT/gen800_sq.json > /dev/null
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r03x/gen800_sq.json'))
for k, v in d.items():
    if 'gen' in k:
        p = v['per_dispatch']; w = p['SQ_WAVES']
        print(k, 'vgpr', v['vgpr'], 'lds', v['lds'], 'grid', v['grid'], 'wg', v['workgroup'])
        print('   per wave: VALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SALU %.0f wave_cycles(quad) %.0f' % (p['SQ_INSTS_VALU']/w, p['SQ_INSTS_LDS']/w, p['SQ_INSTS_VMEM_RD']/w, p['SQ_INSTS_VMEM_WR']/w, p['SQ_INSTS_SALU']/w, p['SQ_WAVE_CYCLES']/w))
        wc = p['SQ_WAVE_CYCLES']
        print('   fractions: active %.3f wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f; bank conflicts/lds_active %.3f' % (p['SQ_ACTIVE_INST_ANY']/wc, p['SQ_WAIT_ANY']/wc, p['SQ_WAIT_INST_ANY']/wc, p['SQ_WAIT_INST_LDS']/wc, p['SQ_LDS_BANK_CONFLICT']/max(p['SQ_LDS_IDX_ACTIVE'],1)))
PY
