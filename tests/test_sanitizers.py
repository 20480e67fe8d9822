"""CPU-only AddressSanitizer + UBSan run of the host-side product coder (csrc/coder_host.cpp over ac_core.h) and of the oracle
(`make -C oracle asan`, driver oracle/asan_driver.cpp): product <-> oracle streams in both directions, degenerate tables, masked
slices, truncated / corrupted streams, ragged conv / table shapes.  GPU sanitizers are not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP host headers")
def test_host_coder_and_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "asan_driver: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
