#!/bin/bash
# GPU box: diagnostic build of cconv16dc_kernels.hip with -DXD_STAMP (+ extra flags) into a scratch copy of the library, plane probe.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/xds.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
objs=$(ls build/*.o | grep -v cconv16dc_kernels.o)
/opt/rocm/bin/hipcc -DXD_STAMP "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -c cconv16dc_kernels.hip -o $T/cconv16dc_kernels.o && \
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv16dc_kernels.o && (cd $R && LIC360_LIB=$T/liblic360_hip.so timeout -k 10 300 python3 tools/xd_plane_probe.py)
