"""Body of __graft_entry__.smoke(): cconv_ec + gmm table on a tiny latent vs the oracle (bit-exact)."""
import numpy as np
import torch

import oracle as orc
from util import conv_params


def run():
    import lic360
    rng = np.random.default_rng(0)
    G, H, W = 6, 8, 16
    w, b, a = conv_params(rng, 3, G * 4, G * 4)
    x = rng.standard_normal((3, G * 4, H, W)).astype(np.float32)
    ref = orc.cconv_ec(x, w, b, a, G, 6)
    op = lic360.CconvEcOp(G * 4, G, G * 4, 5, 6, 0, False)
    d = lambda t: torch.from_numpy(t).to("cuda:0")
    got = op.forward_act_batch(d(x), d(w), d(b), d(a))[0].cpu().numpy()
    assert np.array_equal(got, ref), "cconv_ec mismatch vs oracle"
