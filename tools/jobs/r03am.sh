#!/bin/bash
# SQ counters of the inverse filter bank's kernels
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03am
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
C2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
C3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc $C1 -d $OUT/sq1 -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 2 > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc $C2 -d $OUT/sq2 -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 2 > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc $C3 -d $OUT/sq3 -o run -- python3 $R/tools/bench_next.py f4_ipfb --reps 2 > $OUT/sq3.log 2>&1; echo "sq3 rc=$?"
tail -3 $OUT/sq3.log
cd $R
python3 tools/rocprof_db.py pmc $OUT/sq1/run_results.db $OUT/sq2/run_results.db $OUT/ipfb_sq.json > /dev/null
[ -f $OUT/sq3/run_results.db ] && python3 tools/rocprof_db.py pmc $OUT/sq3/run_results.db $OUT/ipfb_sq3.json > /dev/null
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r03am/ipfb_sq.json'))
for k, v in d.items():
    p = v['per_dispatch']; w = p['SQ_WAVES']; wc = p['SQ_WAVE_CYCLES']
    print(k, 'vgpr', v['vgpr'], 'lds', v['lds'], 'grid', v['grid'], 'wg', v['workgroup'])
    print('   per wave: VALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SALU %.0f wave_cycles %.0f busy_cycles %.0f' % (p['SQ_INSTS_VALU']/w, p['SQ_INSTS_LDS']/w, p['SQ_INSTS_VMEM_RD']/w, p['SQ_INSTS_VMEM_WR']/w, p['SQ_INSTS_SALU']/w, wc/w, p['SQ_BUSY_CYCLES']))
    print('   fractions: active %.3f wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f; bank conflicts/lds_active %.3f' % (p['SQ_ACTIVE_INST_ANY']/wc, p['SQ_WAIT_ANY']/wc, p['SQ_WAIT_INST_ANY']/wc, p['SQ_WAIT_INST_LDS']/wc, p['SQ_LDS_BANK_CONFLICT']/max(p['SQ_LDS_IDX_ACTIVE'],1)))
try:
    d3 = json.load(open('gpurun_out/r03am/ipfb_sq3.json'))
    for k, v in d3.items():
        print(k, {c: round(x, 1) for c, x in v['per_dispatch'].items()})
except Exception as e:
    print('sq3', e)
PY
