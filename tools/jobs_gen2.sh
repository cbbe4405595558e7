#!/bin/bash
# dev: A/B of the generic-length engines on the GPU box (tools/gen2_bench.hip)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG:-gen2}
mkdir -p $OUT
cd $R
run () { echo "== $*" | tee -a $OUT/log.txt; timeout -k 10 120 baseband-tasks_amd/lib/gen2_bench "$@" 2>&1 | tee -a $OUT/log.txt; }
for spec in "${@:-row 490 3402}"; do run $spec || { echo "FAILED: $spec" | tee -a $OUT/log.txt; exit 1; }; done
