#!/bin/bash
# Dev build of libbbt_hip.so with resource report: tools/build.sh [extra hipcc flags] [-- grep-pattern]
set -e
PKG=/root/repo/baseband-tasks_amd
OUT=${BBT_OUT:-$PKG/lib/libbbt_hip.so}
mkdir -p $PKG/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" \
    -o $OUT $PKG/csrc/bbt_hip.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/bbt_build.log || { grep -E "error" -A5 /tmp/bbt_build.log | head -40; exit 1; }
python3 - <<'PY'
import re
txt = open('/tmp/bbt_build.log').read()
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", txt, re.S):
    name = m.group(1)
    short = re.sub(r"^_ZN3bbt\d+", "", name)[:40]
    print(f"{short:42s} vgpr={m.group(2):>3s} occ={m.group(3)} spill={m.group(4)} lds={m.group(5)}")
PY
