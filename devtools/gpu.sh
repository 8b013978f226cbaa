#!/bin/bash
# Build libvdet_hip.so + the oracle HERE (hipcc cross-compiles), then run a script / command on a GPU box:
#     devtools/gpu.sh [--timeout S] 'command'
# (a stale .so travelling to the box cost one GPU call in round 2: never call gpurun directly)
set -e
cd "$(dirname "$0")/.."
bash vdetlib_amd/csrc/build.sh > /dev/null
make -C oracle -B libvdet_oracle.so > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o devtools/valu_bench devtools/valu_bench.hip 2>/dev/null || true
T=1800
if [ "$1" == "--timeout" ]; then T=$2; shift 2; fi
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
