#!/bin/bash
# Runs on the GPU box: rebuild csrc/cconv16dc_kernels.hip with each given flag set INTO A SCRATCH COPY of the library (LIC360_LIB points the
# shim at it; the in-tree liblic360_hip.so is never touched), time the decode probe.
# usage: [LIC360_DC=16] tools/xd_variants.sh "name1:-DFLAG ..." "name2:..."      (results: gpurun_out/xdv_<name>.log)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/xdv.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
objs=$(ls build/*.o | grep -v cconv16dc_kernels.o)
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Rpass-analysis=kernel-resource-usage -c cconv16dc_kernels.hip -o $T/cconv16dc_kernels.o 2> $T/cc.txt
  grep -A12 "k_cconv16dc" $T/cc.txt | grep -E "VGPRs:|Spill|error" | sed "s/.*remark: *//" | tr '\n' ' ' > $R/gpurun_out/xdv_$name.log; echo >> $R/gpurun_out/xdv_$name.log
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv16dc_kernels.o
  (cd $R && LIC360_LIB=$T/liblic360_hip.so PB=${PB:-48} timeout -k 10 300 python3 tools/dc_probe.py 2>&1 | grep -E "dc_hidden|dc_last|exact" >> gpurun_out/xdv_$name.log)
  echo "== $name ($flags)"; cat $R/gpurun_out/xdv_$name.log
done
