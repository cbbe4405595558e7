cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_job.sh sq TAG=r05sq NAME=pfb16 CMD="python3 $GRAFT_REPO_ROOT/tools/bench_next.py pfb_16x4096 --reps 3" 2>&1 | tail -2
bash tools/gpu_job.sh sq TAG=r05sq NAME=ipfb CMD="python3 $GRAFT_REPO_ROOT/tools/bench_next.py f4_ipfb --reps 3" 2>&1 | tail -2
for n in pfb16 ipfb; do python3 tools/rocprof_db.py pmc gpurun_out/r05sq/prof/$n/sq1/run_results.db gpurun_out/r05sq/prof/$n/sq2/run_results.db gpurun_out/r05sq/$n.json > /dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/r05sq/$n.json'))
for k,v in d.items():
    pd=v['per_dispatch']
    if pd.get('SQ_WAVE_CYCLES',0) > 1e6: print('$n', k, 'vgpr',v.get('vgpr'),'lds',v.get('lds'),'wg',v.get('workgroup'),'n',v.get('dispatches'),'wait_any %.3f' % (pd['SQ_WAIT_ANY']/pd['SQ_WAVE_CYCLES']), 'wait_inst %.3f' % (pd['SQ_WAIT_INST_ANY']/pd['SQ_WAVE_CYCLES']),'lds_wait %.3f' % (pd['SQ_WAIT_INST_LDS']/pd['SQ_WAVE_CYCLES']), 'active_valu %.3f' % (pd['SQ_ACTIVE_INST_VALU']/pd['SQ_WAVE_CYCLES']), 'valu/wave %.0f' % (pd['SQ_INSTS_VALU']/pd['SQ_WAVES']), 'vmem_rd/wave %.0f' % (pd['SQ_INSTS_VMEM_RD']/pd['SQ_WAVES']),'lds/wave %.0f' % (pd['SQ_INSTS_LDS']/pd['SQ_WAVES']), 'conflict %.3f' % (pd['SQ_LDS_BANK_CONFLICT']/max(pd['SQ_LDS_IDX_ACTIVE'],1)))
"; done
timeout -k 10 300 python3 tools/bench_next.py pfb_16x4096 f4_ipfb pfb_4x1024 2>/dev/null
