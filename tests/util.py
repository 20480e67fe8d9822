"""Shared helpers for the parity tests: seeded synthetic weights / latents."""
import numpy as np


def rng_for(*key):
    return np.random.default_rng(abs(hash(tuple(key))) % (2 ** 32))


def conv_params(rng, nb, nout, C, k=5, act=True):
    """Kaiming-normal weights, bias U(-.1,.1), PReLU slope 0.25 (+ jitter so that it matters)."""
    fan_in = C * k * k
    lead = () if nb is None else (nb,)
    w = (rng.standard_normal(lead + (nout, C, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, lead + (nout,)).astype(np.float32)
    a = (0.25 + rng.uniform(-0.05, 0.05, lead + (nout,))).astype(np.float32) if act else None
    return w, b, a


def latent(rng, G, H, W):
    """code in {0..7} ~ round(N(3.5,1.2^2)); importance level L[h/2,w/2] in {0..G}; mask[g,y,x] = g < L."""
    code = np.clip(np.rint(rng.normal(3.5, 1.2, (1, G, H, W))), 0, 7).astype(np.float32)
    L = np.clip(np.rint(G / 2 + G / 4 * rng.standard_normal((H // 2, W // 2))), 0, G).astype(np.int64)
    Lup = np.repeat(np.repeat(L, 2, 0), 2, 1)
    mask = (np.arange(G)[:, None, None] < Lup[None]).astype(np.float32)[None]
    return code, mask, L.astype(np.float32)[None, None]
