#!/bin/bash
# the documented switches still give correct results: the parity file under each alternative setting
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03bb
mkdir -p $OUT
run () { name=$1; shift; env "$@" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q > $OUT/$name.log 2>&1; echo "$name rc=$? $(tail -1 $OUT/$name.log)"; }
run tw_col0 BBT_OSM_TW_COL=0
run tw4_tables0 BBT_OSM_TW4_TABLES=0
run planar0_wide0 BBT_PLANAR=0 BBT_COL_WIDE=0 BBT_COL_LDS_PAD_FIRST=0
run hostpipe0_short0 BBT_HOST_PIPELINE=0 BBT_SHORT_BLOCK=0
run fuse_dechan1_fuse0 BBT_FUSE_DECHANNELIZE=1 BBT_FUSE=0
