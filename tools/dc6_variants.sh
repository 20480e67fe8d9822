#!/bin/bash
# GPU box: rebuild csrc/cconv4_kernels.hip with each given flag set into a scratch copy of the library and time the decode-order 4x4x1 kernel on the
# probe planes (tools/dc_plane_probe.py).  usage: [XP=...] tools/dc6_variants.sh "name1:-DFLAG ..." "name2:..."
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/d6v.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Rpass-analysis=kernel-resource-usage -c cconv4_kernels.hip -o $T/cconv4_kernels.o 2> $T/cc.txt || { tail -20 $T/cc.txt; continue; }
  echo "== $name ($flags): $(grep -A8 'k_cconv4v6ILi4ELb0ELb0EE' $T/cc.txt | grep -E ' VGPRs:|VGPRs Spill' | sed 's/.*remark: *//' | tr '\n' ' ')"
  objs=$(ls build/*.o | grep -v cconv4_kernels.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv4_kernels.o
  (cd $R && LIC360_LIB=$T/liblic360_hip.so timeout -k 10 300 python3 tools/dc_plane_probe.py 2>&1 | grep -E "^plane|^mean")
done
