#!/bin/bash
# GPU box: diagnostic build of cconv4_kernels.hip with -DDC6_STAMP (+ extra flags) into a scratch copy of the library, plane probe
# (cycles per phase and wave of the decode-order 4x4x1 hidden-layer kernel).  usage: [XP=150 XWAVES=1] tools/dc6_stamp.sh [-DFLAG ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/d6s.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
objs=$(ls build/*.o | grep -v cconv4_kernels.o)
/opt/rocm/bin/hipcc -DDC6_STAMP "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -c cconv4_kernels.hip -o $T/cconv4_kernels.o 2> $T/cc.txt || { tail -20 $T/cc.txt; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv4_kernels.o && (cd $R && LIC360_LIB=$T/liblic360_hip.so timeout -k 10 300 python3 tools/dc_plane_probe.py)
