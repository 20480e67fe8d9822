#!/bin/bash
# GPU box: rebuild csrc/cconv4_kernels.hip with each given flag set into a scratch copy of the library and time ONE decode of PB images through the fused
# codec (tools/dc_probe.py: the dead-cone LIST kernels on SURVEY 8d's masks).  usage: [PB=64] tools/dc6_list_variants.sh "name1:-DFLAG ..." "name2:..."
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d /tmp/d6l.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd $R/360-image-compression_amd/csrc
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Rpass-analysis=kernel-resource-usage -c cconv4_kernels.hip -o $T/cconv4_kernels.o 2> $T/cc.txt || { tail -20 $T/cc.txt; continue; }
  echo "== $name ($flags): $(grep -A8 'k_cconv4v6lILi4' $T/cc.txt | grep -E ' VGPRs:|ScratchSize' | sed 's/.*remark: *//' | tr '\n' ' ')"
  objs=$(ls build/*.o | grep -v cconv4_kernels.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/liblic360_hip.so $objs $T/cconv4_kernels.o
  for rep in 1 2; do (cd $R && LIC360_LIB=$T/liblic360_hip.so PB=${PB:-64} timeout -k 10 300 python3 tools/dc_probe.py 2>&1 | grep -E "dc_hidden|dc_last|wall" | tr '\n' ' '; echo); done
done
