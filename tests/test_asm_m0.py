"""M0 guard: the coder kernels of codec_fused.hip and the LDS-DMA helpers write M0 inside inline-asm statements
without listing it as a clobber (hipcc treats M0 as reserved and only warns about such a clobber).  That is safe exactly as long as the
COMPILER's own code never reads M0 in those kernels.  This test compiles the device side to assembly and checks it: every instruction
that mentions m0 lies between ;;#ASMSTART and ;;#ASMEND.  A future hipcc that starts using M0 there fails here, not on the GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "360-image-compression_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["codec_fused.hip", "cconv16dc_kernels.hip"])
def test_m0_only_inside_inline_asm(tmp_path, src):
    out = str(tmp_path / "k.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
                           "-c", os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    inside, bad, seen = False, [], 0
    for line in open(out):
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        code = line.split(";")[0]
        if re.search(r"\bm0\b", code):
            seen += 1
            if not inside:
                bad.append(line.strip())
    assert seen > 0, "no M0 use found at all: the test no longer sees the asm statements"
    assert not bad, "compiler-generated M0 use next to asm statements that overwrite M0: %s" % bad[:5]
