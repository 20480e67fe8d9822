"""The dead cone against the ORACLE's own arithmetic (no GPU): run the 12 layers of the entropy nets with orc.cconv_ec (the restatement of
extension/cconv_ec_cuda.cu:268-315), and between every two layers overwrite every DEAD cell -- group above need_l at its position, need from the
brute-force reachability statement of tests/test_gpu_need.py -- with a huge finite value.  The last layer's outputs at every CODED symbol must not move
by one bit: what the GPU codec's skip relies on, shown with the reference's rule for which inputs a chain reads.  (A finite poison, not NaN: the dense
oracle multiplies nothing by zero weights, but the GPU kernels do, and the statement to prove is about non-zero weights.)"""
import numpy as np
import pytest

import oracle as orc
import ref_codec as rc
from test_gpu_need import brute_need
from util import latent, latent_smooth


def _forward(x, layers, G, need=None, poison=None):
    """net_ec of ref_codec.py with the dead cells of every layer's output poisoned before the next layer reads them"""
    def kill(y, l, cpg):
        if need is None:
            return y
        y = y.copy()
        g = np.arange(G)[:, None, None]
        dead = g > need[l][None]                                            # [G, H, W]
        y.reshape(y.shape[0], G, cpg, *y.shape[2:])[:, dead.reshape(G, *y.shape[2:])[:, None].repeat(cpg, 1)] = poison
        return y
    y = kill(orc.cconv_ec(x, layers[0]["w"], layers[0]["b"], layers[0]["a"], G, layers[0]["constrain"]), 0, 4)
    for i in range(5):
        l1, l2 = layers[1 + 2 * i], layers[2 + 2 * i]
        t = kill(orc.cconv_ec(y, l1["w"], l1["b"], l1["a"], G, 6), 1 + 2 * i, 4)
        t = orc.cconv_ec(t, l2["w"], l2["b"], l2["a"], G, 6)
        y = kill(t + y, 2 + 2 * i, 4)                                       # (the residual's own dead cells are dead in the sum: need_{l-2} >= need_l at a position)
    l = layers[11]
    return orc.cconv_ec(y, l["w"], l["b"], l["a"], G, 6)


@pytest.mark.parametrize("G,H,W,kind,seed", [(6, 8, 12, "iid", 1), (12, 8, 10, "smooth", 2), (9, 6, 6, "blob", 3)])
def test_poisoned_dead_cells_do_not_reach_a_coded_symbol(G, H, W, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "smooth":
        code, mask, _ = latent_smooth(rng, G, H, W)
    elif kind == "iid":
        code, mask, _ = latent(rng, G, H, W)
    else:
        code, _, _ = latent(rng, G, H, W)
        L = np.zeros((H, W), np.int64)
        L[2:4, 1:4] = G - 2
        mask = (np.arange(G)[:, None, None] < L[None]).astype(np.float32)[None]
    layers = rc.make_main_params(500 + seed, G)
    need = brute_need(mask[0] > 0.5)
    t = ((code - np.float32(3.5)) * mask).astype(np.float32)
    x = np.concatenate([t, t, t], 0)
    want = _forward(x, layers, G)
    coded = (mask[0] > 0.5)[:, None].repeat(3, 1).reshape(3 * G, H, W)       # the last layer's three channels of a coded symbol
    for poison in (np.float32(1e10), np.float32(-3e8)):
        got = _forward(x, layers, G, need, poison)
        assert np.array_equal(got[:, coded], want[:, coded])
    assert (need[11] < G - 1).any()                                          # (there was something dead to poison)
    # ... and the cone is tight somewhere: poisoning one group BELOW need at the layer under the last one does change a coded output
    if (need[10] >= 0).any():
        tight = [need[l].copy() for l in range(12)]
        tight[10] = need[10] - 1
        assert not np.array_equal(_forward(x, layers, G, tight, np.float32(1e10))[:, coded], want[:, coded])
