#!/bin/bash
# Runs on the GPU box: rebuild csrc/cconv16_kernels.hip with each given flag set INTO A SCRATCH COPY of the library (the in-tree
# liblic360_hip.so is never touched: LIC360_LIB points the shim at the copy), time the encode probe.
# usage: tools/ec_variants.sh "name1:-DFLAG ..." "name2:..."      (results: gpurun_out/ecv_<name>.log)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/ecv.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Rpass-analysis=kernel-resource-usage -c cconv16_kernels.hip -o $T/cconv16_kernels.o 2> $T/cc.txt || { tail -20 $T/cc.txt; rm -f $T/cconv16_kernels.o; continue; }
  grep -A12 "k_cconv16ILi4ELb0" $T/cc.txt | grep -E "VGPRs:|VGPRs Spill|error" | sed "s/.*remark: *//" | tr '\n' ' ' > $R/gpurun_out/ecv_$name.log; echo >> $R/gpurun_out/ecv_$name.log
  objs=$(ls build/*.o | grep -v cconv16_kernels.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv16_kernels.o
  rm -f $T/cconv16_kernels.o
  (cd $R && LIC360_LIB=$T/liblic360_hip.so PB=${PB:-48} timeout -k 10 300 python3 tools/ec_probe.py 2>&1 | grep -E "ec_|rror" >> gpurun_out/ecv_$name.log)
  echo "== $name ($flags)"; cat $R/gpurun_out/ecv_$name.log
done
