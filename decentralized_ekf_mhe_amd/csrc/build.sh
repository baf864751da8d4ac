#!/bin/bash
# Builds libdekf.so (the C-ABI shared library, gfx950 only) in-tree.
#   DEKF_OUT=libdekf_prof.so DEKF_UNITY=1 bash build.sh -DDEKF_PROFILE     diagnostic / A-B variants under another name
# DEKF_UNITY=1 compiles both sources as ONE translation unit without -fgpu-rdc (no link-time code generation): the
# -DDEKF_PROFILE variant trips a code-generator bug of this ROCm in the LTO step ("Illegal instruction detected:
# V_CMP_NE_U32_e32 0, $src_shared_base"); the unity build does not.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${DEKF_OUT:-libdekf.so}
if [ -n "$DEKF_UNITY" ]; then
    U=$(mktemp /tmp/dekf_unity_XXXXXX.hip)
    printf '#include "%s/kernels.hip"\n#include "%s/dekf_capi.hip"\n' "$(pwd)" "$(pwd)" > "$U"
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I/opt/rocm/include -I"$(pwd)" -o "$OUT" "$U" -ldl "$@"
    rm -f "$U"
else
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fgpu-rdc \
        -I/opt/rocm/include -o "$OUT" kernels.hip dekf_capi.hip -ldl "$@"
fi
echo "built $(pwd)/$OUT"
