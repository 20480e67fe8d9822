"""The synthetic workload generators are pinned by digests inside the committed goldens (a drifting generator must fail here, on the CPU, not as a
byte mismatch on the GPU box): SURVEY.md 8d's smooth importance maps (round 6) and the i.i.d. maps of rounds 1-5."""
import hashlib
import os

import numpy as np
import pytest

from util import make_latent, latent_smooth, smooth_field

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["cfg2", "cfg3b", "cfg2s", "cfg3s", "cfg5s"])
def test_generators_reproduce_the_goldens_inputs(name):
    g = np.load(os.path.join(GOLD, "full_%s.npz" % name))
    kind = str(g["kind"]) if "kind" in g.files else "iid"
    dens = (float(g["mean"]), float(g["spread"])) if "mean" in g.files else (0.5, 0.25)
    code, mask, _ = make_latent(kind, np.random.default_rng(int(g["latent_seed"])), int(g["G"]), int(g["H"]), int(g["W"]), *dens)
    assert hashlib.sha256(code.tobytes()).hexdigest() == str(g["code_sha256"])
    assert hashlib.sha256(mask.tobytes()).hexdigest() == str(g["mask_sha256"])


def test_smooth_maps_are_what_survey_8d_asks_for():
    """L = clip(round(24 + 12 cos(lat) n)), n smooth and of unit variance: flat (24) at the poles, +-12 n at the equator, a prefix mask in g"""
    f = np.stack([smooth_field(np.random.default_rng(s), 32, 64) for s in range(64)])
    assert 0.9 < f[:, 8:24].std() < 1.1                                      # unit variance away from the reflecting poles
    assert np.corrcoef(f[:, 16, :-1].ravel(), f[:, 16, 1:].ravel())[0, 1] > 0.9   # smooth: neighbouring cells move together (sigma = 3 cells)
    code, mask, lv = latent_smooth(np.random.default_rng(7), 48, 64, 128)
    L = lv[0, 0]
    assert np.all(np.abs(L[0] - 24) <= 2) and np.all(np.abs(L[-1] - 24) <= 2) and L[12:20].std() > 6
    assert np.array_equal(mask[0].sum(0), np.repeat(np.repeat(L, 2, 0), 2, 1))   # g < L[y/2, x/2]
    assert set(np.unique(code)) <= set(range(8)) and 0.3 < mask.mean() < 0.7
