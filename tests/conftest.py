import os
import sys

# Before numpy / torch / the oracle load their thread pools: a GPU box reports 256 hardware threads, and three pools of that size in one
# process (OpenMP in the oracle, the BLAS behind numpy, torch's intra-op pool) reach the box's per-job thread limit -- libgomp aborts when a
# thread cannot be created (seen once in round 5 as "Fatal Python error: Aborted" inside orc_cconv_ec) -- and forking 256 threads for the
# small planes of the op-level driver tests made them take minutes.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(min(32, os.cpu_count() or 1)))

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "360-image-compression_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ac_golden.npz"))
