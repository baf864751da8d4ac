#!/bin/bash
# Builds libdekf.so (the C-ABI shared library, gfx950 only) in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fgpu-rdc \
    -I/opt/rocm/include -o libdekf.so kernels.hip dekf_capi.hip -ldl "$@"
echo "built $(pwd)/libdekf.so"
