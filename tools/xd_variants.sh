#!/bin/bash
# Runs on the GPU box: rebuild csrc/cconv16dc_kernels.hip with each given flag set, relink the library, time the decode probe.
# usage: tools/xd_variants.sh "name1:-DFLAG ..." "name2:..."      (results: gpurun_out/xdv_<name>.log; the in-tree .so is restored)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/360-image-compression_amd/csrc
cp ../liblic360_hip.so /tmp/liblic360_hip.so.orig
cp build/cconv16dc_kernels.o /tmp/cconv16dc_kernels.o.orig
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Rpass-analysis=kernel-resource-usage -c cconv16dc_kernels.hip -o build/cconv16dc_kernels.o 2> /tmp/xdv_$name.cc
  grep -A12 "k_cconv16dc" /tmp/xdv_$name.cc | grep -E "VGPRs:|Spill|error" | tr '\n' ' ' > $R/gpurun_out/xdv_$name.log; echo >> $R/gpurun_out/xdv_$name.log
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../liblic360_hip.so build/*.o
  (cd $R && PB=${PB:-48} timeout -k 10 300 python3 tools/dc_probe.py 2>&1 | grep -E "dc_hidden|dc_last|exact" >> gpurun_out/xdv_$name.log)
  echo "== $name ($flags)"; cat $R/gpurun_out/xdv_$name.log
done
cp /tmp/liblic360_hip.so.orig ../liblic360_hip.so
cp /tmp/cconv16dc_kernels.o.orig build/cconv16dc_kernels.o
