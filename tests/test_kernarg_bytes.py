"""The kernels that are launched once per decode plane keep their kernel arguments small and carry no implicit (hidden) arguments.

Why this is a test: on MI355X the three-stream throughput step and the dependent-launch latency both pay for every byte of kernel arguments
(DESIGN.md 4.1 b'', profiles/r05_dc6_micro_variants.txt: padding the decode kernel's struct from 128 to 272 bytes cost the 1024x2048
configuration 6 %; the 256 bytes of implicit arguments a HIP kernel gets as soon as it reads gridDim / blockDim cost the headline 2 % and a
single-image decode 2 us per launch).  The budget is checked on the BUILT library: the gfx950 code objects are carved out of the clang
offload bundles in liblic360_hip.so and their AMDGPU metadata is read with llvm-readelf.  No GPU needed."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "360-image-compression_amd", "liblic360_hip.so")
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"

# substring of the mangled kernel name -> largest kernarg segment it may have (explicit bytes; any hidden argument adds 256 and fails)
BUDGET = {
    "k_cconv4v6ILi": 96,            # decode-order conv of the latent nets: Dc3Packed
    "k_cconv4v6tILi": 160,          # ... taped launches: + Dc3Tape
    "k_cconv144ILi": 160,           # importance-map net, both orders: I144Args
    "12k_dec_tables": 96,
    "11k_dec_planeILb": 128,
    "k_imp_dec_tablesILb": 64,
    "k_imp_dec_planeILb": 128,
    "16k_imp_mask_plane": 64,
    "10k_cconv_dc": 208,            # generic decode-order kernel (first layer of the importance net)
}


def _kernels():
    blob = open(LIB, "rb").read()
    out = {}
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        p = m.start()
        (nent,) = struct.unpack_from("<Q", blob, p + 24)
        off = p + 32
        for _ in range(nent):
            eo, es, ts = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + ts].decode()
            off += ts
            if "gfx950" not in triple or not es:
                continue
            path = "/tmp/lic360_co_%d.elf" % os.getpid()
            with open(path, "wb") as f:
                f.write(blob[p + eo:p + eo + es])
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
            os.unlink(path)
            for block in notes.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                size = re.search(r"\.kernarg_segment_size:\s+(\d+)", block)
                if name and size:
                    out[name.group(1)] = (int(size.group(1)), "hidden_" in block)
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="built library or llvm-readelf missing")
def test_per_plane_kernels_have_small_arguments_and_no_hidden_ones():
    ks = _kernels()
    assert len(ks) > 100, "code objects of the library not found"
    seen = set()
    for name, (size, hidden) in ks.items():
        for pat, budget in BUDGET.items():
            if pat in name:
                seen.add(pat)
                assert not hidden, "%s reads gridDim / blockDim (or another implicit argument): +256 bytes per launch" % name
                assert size <= budget, "%s: %d bytes of kernel arguments, budget %d" % (name, size, budget)
    assert seen == set(BUDGET), "kernels not found in the library: %s" % sorted(set(BUDGET) - seen)
