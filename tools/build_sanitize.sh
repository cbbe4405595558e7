#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libbbt_hip.so:
# the plan / pool / descriptor logic of csrc/bbt_hip.hip instrumented on the
# CPU, device code compiled as usual (GPU ASan is not available on this pool).
#   tools/build_sanitize.sh            -> build/libbbt_hip_asan.so
# Use:  LD_PRELOAD=$(tools/build_sanitize.sh --runtime) ASAN_OPTIONS=detect_leaks=0 \
#       BBT_HIP_LIB=$PWD/build/libbbt_hip_asan.so python -m pytest tests/test_cabi.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANG=/opt/rocm/lib/llvm/bin/clang
if [ "$1" = "--runtime" ]; then
    $CLANG -print-file-name=libclang_rt.asan-x86_64.so
    exit 0
fi
mkdir -p "$ROOT/build"
$HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared \
    -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-value \
    -o "$ROOT/build/libbbt_hip_asan.so" "$ROOT/baseband-tasks_amd/csrc/bbt_hip.hip"
echo "$ROOT/build/libbbt_hip_asan.so"
