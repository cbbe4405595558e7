#!/bin/bash
# headline A/B: lanes x chunk, two LDS regions in the row pass
OUT=$GRAFT_REPO_ROOT/gpurun_out/sweep
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
one () { local tag=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --no-cpu --no-verify --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['roofline']['pass_ms_per_block'])" | tee -a $OUT/sweep.txt
}
one base BBT_X=0
one lanes3_chunk4 BBT_OSM_LANES=3 BBT_OSM_CHUNK=4
one lanes2_chunk4 BBT_OSM_LANES=2 BBT_OSM_CHUNK=4
one lanes2_chunk8 BBT_OSM_LANES=2 BBT_OSM_CHUNK=8
one lanes3_chunk6 BBT_OSM_LANES=3 BBT_OSM_CHUNK=6
one lanes4_chunk3 BBT_OSM_LANES=4 BBT_OSM_CHUNK=3
one two_regions BBT_HIP_LIB=$PWD/build/libbbt_hip_2reg.so
one two_regions_lanes3_chunk4 BBT_HIP_LIB=$PWD/build/libbbt_hip_2reg.so BBT_OSM_LANES=3 BBT_OSM_CHUNK=4
