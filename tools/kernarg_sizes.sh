#!/bin/bash
# Kernel-argument bytes of every kernel of the library (explicit arguments + HIP's 256 bytes of implicit ones where a kernel reads gridDim / blockDim or
# calls something that needs them).  Argument bytes cost time on this machine when several streams launch at once ("argument bytes"):
#   tools/kernarg_sizes.sh [file.hip ...]
cd "$(dirname "$0")/../360-image-compression_amd/csrc" || exit 1
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden --cuda-device-only -S"
for f in ${@:-*.hip}; do
  case $f in *.hip) x="" ;; *) x="-x hip" ;; esac
  /opt/rocm/bin/hipcc $F $x $f -o /tmp/ks_$$.s 2>/dev/null || { echo "== $f: compile failed"; continue; }
  echo "== $f"
  awk '/^[ \t]*\.amdhsa_kernel /{n=$2} /\.amdhsa_kernarg_size/{print $2, n}' /tmp/ks_$$.s | sort -n | c++filt | cut -c1-150
done
rm -f /tmp/ks_$$.s
