"""M0 guard (every translation unit that writes M0: ADVICE r4): the coder kernels of codec_fused.hip and the LDS-DMA helpers write M0 inside inline-asm statements
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


def _writes_m0(name):
    """does this translation unit, or a csrc header / include it pulls in, contain an `s_mov_b32 m0` statement?"""
    text = open(os.path.join(CSRC, name)).read()
    parts = [text] + [open(os.path.join(CSRC, i)).read() for i in os.listdir(CSRC) if i.endswith((".inc", ".h")) and ('"%s"' % i) in text]
    return any(re.search(r"s_mov_b32\s+m0", t) for t in parts)


M0_WRITERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and _writes_m0(f))


def test_the_m0_writing_sources_are_the_known_ones():
    assert M0_WRITERS == ["cconv144_kernels.hip", "cconv16_kernels.hip", "codec_fused.hip", "conv3x3_kernels.hip"], M0_WRITERS


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", M0_WRITERS)
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
