cd $GRAFT_REPO_ROOT
bash tools/gpu_job.sh ab TAG=opsel A=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib/libbbt_hip_noopsel.so B=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib/libbbt_hip_opsel.so N=3
for v in noopsel opsel; do for c in config1 config3; do BBT_HIP_LIB=$GRAFT_REPO_ROOT/baseband-tasks_amd/lib/libbbt_hip_$v.so timeout -k 10 200 python3 tools/bench_one.py $c 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v $c', d.get('msamples_per_s'))"; done; done
