#!/bin/bash
# Build libvdet_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: the f32 operation order of the IoU predicate is the specification.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
    -Wall -Wno-unused-result \
    -o ../libvdet_hip.so vdet_capi.hip "$@"
echo "built $(cd .. && pwd)/libvdet_hip.so"
# host-side marshalling helper (plain C, CPython API): protocol dicts -> float32 rows
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
${CC:-gcc} -O2 -fPIC -shared -Wall -I"$PYINC" -o ../_protofast.so protofast.c
echo "built $(cd .. && pwd)/_protofast.so"
