"""f1 oracle restatements (oracle.blocks: the reference's ResidualBlock / V2 / Down / Up, test/model_zoo.py:8-95,145-170, over orc_conv2d /
orc_prelu / orc_gdn and the sphere-op restatements) against index-only torch CPU statements of the same blocks: independent code, fp32,
1e-5.  The -m gpu test tests/test_gpu_models.py::test_blocks_match_the_oracle compares lic360_models.py with these restatements."""
import numpy as np
import torch
import torch.nn.functional as F

import oracle as orc


def _pad(x, pad):
    W = x.shape[-1]
    body = torch.cat([x[..., W - pad:], x, x[..., :pad]], -1)

    def across(r):
        r = torch.flip(r, (-1,))
        return torch.cat([r[..., W - pad:], r, r[..., :pad]], -1)
    return torch.cat([across(torch.flip(x[..., :pad, :], (-2,))), body, across(torch.flip(x[..., x.shape[-2] - pad:, :], (-2,)))], -2)


def _refresh(x):
    return _pad(x[..., 2:-2, 2:-2], 2)


def _trim(x, p):
    y = torch.zeros_like(x)
    y[..., p:-p, p:-p] = x[..., p:-p, p:-p]
    return y


def _gdn_consts(c, g):
    ped, gb = 2.0 ** -36, 2.0 ** -18
    bb = (1e-6 + ped) ** 0.5
    gamma = torch.sqrt(0.1 * torch.eye(c) + ped) + 0.01 * torch.rand((c, c), generator=g)
    beta = torch.sqrt(torch.ones(c) + ped) + 0.1 * torch.rand((c,), generator=g)
    ge, be = torch.clamp(gamma, min=gb) ** 2 - ped, torch.clamp(beta, min=bb) ** 2 - ped
    return gamma, beta, ge, be, {"relu2.gamma": gamma.numpy(), "relu2.beta": beta.numpy(), "relu2.pedestal": ped, "relu2.beta_bound": bb, "relu2.gamma_bound": gb}


def _conv(g, co, ci, k):
    return torch.randn((co, ci, k, k), generator=g) * 0.1, torch.randn((co,), generator=g) * 0.1


def test_oracle_blocks_equal_torch_statements():
    g = torch.Generator().manual_seed(0)
    c = 8
    x = _refresh(torch.randn((2, c, 10, 14), generator=g))
    xr = _refresh(x)
    # ---- ResidualBlock
    (w1, b1), (w2, b2), (w3, b3) = _conv(g, c // 2, c, 1), _conv(g, c // 2, c // 2, 3), _conv(g, c, c // 2, 1)
    a1, a2 = torch.rand(c // 2, generator=g) * 0.3, torch.rand(c // 2, generator=g) * 0.3
    want = _trim(xr + F.conv2d(F.prelu(F.conv2d(F.prelu(F.conv2d(xr, w1, b1), a1), w2, b2, 1, 1), a2), w3, b3), 2)
    p = {"conv1.weight": w1.numpy(), "conv1.bias": b1.numpy(), "relu1.weight": a1.numpy(), "conv2.weight": w2.numpy(), "conv2.bias": b2.numpy(),
         "relu2.weight": a2.numpy(), "conv3.weight": w3.numpy(), "conv3.bias": b3.numpy()}
    assert np.allclose(orc.blocks.residual(x.numpy().copy(), p), want.numpy(), rtol=1e-5, atol=1e-5)
    # ---- ResidualBlockV2
    (w1, b1), (w2, b2) = _conv(g, c, c, 3), _conv(g, c, c, 3)
    a1, a2 = torch.rand(c, generator=g) * 0.3, torch.rand(c, generator=g) * 0.3
    y = _trim(F.prelu(F.conv2d(xr, w1, b1, 1, 1), a1), 1)
    want = xr + _trim(F.prelu(F.conv2d(y, w2, b2, 1, 1), a2), 2)
    p = {"conv1.weight": w1.numpy(), "conv1.bias": b1.numpy(), "relu1.weight": a1.numpy(), "conv2.weight": w2.numpy(), "conv2.bias": b2.numpy(),
         "relu2.weight": a2.numpy()}
    assert np.allclose(orc.blocks.residual_v2(x.numpy().copy(), p), want.numpy(), rtol=1e-5, atol=1e-5)
    # ---- ResidualBlockDown (hidden)
    (w1, b1), (w2, b2), (ws, bs) = _conv(g, c, c, 3), _conv(g, c, c, 3), _conv(g, c, c, 1)
    a1 = torch.rand(c, generator=g) * 0.3
    gamma, beta, ge, be, gp = _gdn_consts(c, g)
    t = F.conv2d(x, ws, bs, 2, 2)
    y = _refresh(_trim(F.prelu(F.conv2d(xr, w1, b1, 2, 3), a1), 2))
    yc = F.conv2d(y, w2, b2, 1, 1)
    want = _trim(t + yc / torch.sqrt(F.conv2d(yc * yc, ge.view(c, c, 1, 1), be)), 2)
    p = dict({"conv1.weight": w1.numpy(), "conv1.bias": b1.numpy(), "relu1.weight": a1.numpy(), "conv2.weight": w2.numpy(), "conv2.bias": b2.numpy(),
              "short_cut.weight": ws.numpy(), "short_cut.bias": bs.numpy()}, **gp)
    assert np.allclose(orc.blocks.residual_down(x.numpy().copy(), p), want.numpy(), rtol=1e-5, atol=1e-5)
    # ---- ResidualBlockUp
    (w1, b1), (w2, b2), (ws, bs) = _conv(g, 4 * c, c, 3), _conv(g, c, c, 3), _conv(g, 4 * c, c, 1)
    a1 = torch.rand(4 * c, generator=g) * 0.3
    gamma, beta, ge, be, gp = _gdn_consts(c, g)
    b = _trim(F.pixel_shuffle(F.prelu(F.conv2d(xr, w1, b1), a1), 2), 2)
    bc = F.conv2d(_refresh(b), w2, b2, 1, 1)
    b = bc * torch.sqrt(F.conv2d(bc * bc, ge.view(c, c, 1, 1), be))
    want = _trim(b + F.pixel_shuffle(F.conv2d(xr[..., 1:-1, 1:-1], ws, bs), 2), 2)
    p = dict({"conv1.weight": w1.numpy(), "conv1.bias": b1.numpy(), "relu1.weight": a1.numpy(), "conv2.weight": w2.numpy(), "conv2.bias": b2.numpy(),
              "short_cut.weight": ws.numpy(), "short_cut.bias": bs.numpy()}, **gp)
    got = orc.blocks.residual_up(x.numpy().copy(), p)
    assert got.shape == tuple(want.shape) and np.allclose(got, want.numpy(), rtol=1e-5, atol=1e-5)
