#!/bin/bash
# GPU box: diagnostic build of cconv16dc_kernels.hip with -DXD_STAMP (+ extra flags), plane probe; restores the in-tree library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/360-image-compression_amd/csrc
cp ../liblic360_hip.so /tmp/liblic360_hip.so.orig; cp build/cconv16dc_kernels.o /tmp/cconv16dc_kernels.o.orig
/opt/rocm/bin/hipcc -DXD_STAMP "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -c cconv16dc_kernels.hip -o build/cconv16dc_kernels.o && \
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../liblic360_hip.so build/*.o && (cd $R && timeout -k 10 300 python3 tools/xd_plane_probe.py)
cp /tmp/liblic360_hip.so.orig ../liblic360_hip.so; cp /tmp/cconv16dc_kernels.o.orig build/cconv16dc_kernels.o
