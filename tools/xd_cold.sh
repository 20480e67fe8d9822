#!/bin/bash
# GPU box: rebuild cconv16dc_kernels.hip with the given flags, run the plane probe with cold inputs (8 rotating buffers); restores the library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/360-image-compression_amd/csrc
cp ../liblic360_hip.so /tmp/liblic360_hip.so.orig; cp build/cconv16dc_kernels.o /tmp/cconv16dc_kernels.o.orig
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -c cconv16dc_kernels.hip -o build/cconv16dc_kernels.o && \
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../liblic360_hip.so build/*.o && (cd $R && XCOLD=${XCOLD:-8} XP=${XP:-20,50,80,110,170,200,224} timeout -k 10 300 python3 tools/xd_plane_probe.py 2>&1 | grep -E "plane|mean" | sed "s/  | wave.*| old/ | old/")
cp /tmp/liblic360_hip.so.orig ../liblic360_hip.so; cp /tmp/cconv16dc_kernels.o.orig build/cconv16dc_kernels.o
