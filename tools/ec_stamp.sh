#!/bin/bash
# GPU box: diagnostic build of cconv16_kernels.hip with -DC16_STAMP (+ extra flags) into a scratch copy of the library, encode probe
# (cycles per phase and wave of the encode-order hidden layer).  usage: tools/ec_stamp.sh [-DFLAG ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/ecs.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
objs=$(ls build/*.o | grep -v cconv16_kernels.o)
/opt/rocm/bin/hipcc -DC16_STAMP "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -c cconv16_kernels.hip -o $T/cconv16_kernels.o 2> $T/cc.txt || { tail -20 $T/cc.txt; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv16_kernels.o && (cd $R && LIC360_LIB=$T/liblic360_hip.so PB=${PB:-48} timeout -k 10 300 python3 tools/ec_probe.py)
