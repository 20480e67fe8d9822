"""CPU check of the full-size fixtures: the seeded inputs regenerate bit for bit (numpy Generator streams are what the
fixtures depend on) and the stored digests match the stored bytes."""
import hashlib
import os

import numpy as np
import pytest

from util import latent

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg5"])
def test_fixture_is_self_consistent(name):
    g = np.load(os.path.join(GOLD, "full_%s.npz" % name))
    code, mask, _ = latent(np.random.default_rng(int(g["latent_seed"])), int(g["G"]), int(g["H"]), int(g["W"]))
    assert hashlib.sha256(code.tobytes()).hexdigest() == str(g["code_sha256"])
    assert hashlib.sha256(mask.tobytes()).hexdigest() == str(g["mask_sha256"])
    assert hashlib.sha256(g["bytes"].tobytes()).hexdigest() == str(g["sha256"])
    assert hashlib.sha256(g["imp_bytes"].tobytes()).hexdigest() == str(g["imp_sha256"])
    # a coded stream of this model is ~1.3 bit per coded symbol: sanity of the size only
    assert 0.05 < 8.0 * len(g["bytes"]) / mask.sum() < 4.0
