"""Shared helpers for the parity tests: seeded synthetic weights / latents."""
import numpy as np


def _stable(key):
    """a process-independent integer of a (nested) tuple of ints / bools / None / strings: hash() of a tuple holding None depends
    on the object's address in CPython 3.10, so seeds drawn from it changed from run to run (VERDICT r2 #3b)"""
    import zlib
    return zlib.crc32(repr(key).encode())


def rng_for(*key):
    return np.random.default_rng(_stable(tuple(key)))


def case_rng(case):
    return np.random.default_rng(_stable(case))


def conv_params(rng, nb, nout, C, k=5, act=True):
    """Kaiming-normal weights, bias U(-.1,.1), PReLU slope 0.25 (+ jitter so that it matters)."""
    fan_in = C * k * k
    lead = () if nb is None else (nb,)
    w = (rng.standard_normal(lead + (nout, C, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, lead + (nout,)).astype(np.float32)
    a = (0.25 + rng.uniform(-0.05, 0.05, lead + (nout,))).astype(np.float32) if act else None
    return w, b, a


def latent(rng, G, H, W, mean=0.5, spread=0.25):
    """code in {0..7} ~ round(N(3.5,1.2^2)); importance level L[h/2,w/2] in {0..G} ~ round(G (mean + spread N(0,1))); mask[g,y,x] = g < L.
    (mean / spread: the mask density; the defaults are the generator every committed digest was made with)"""
    code = np.clip(np.rint(rng.normal(3.5, 1.2, (1, G, H, W))), 0, 7).astype(np.float32)
    L = np.clip(np.rint(G * mean + G * spread * rng.standard_normal((H // 2, W // 2))), 0, G).astype(np.int64)
    Lup = np.repeat(np.repeat(L, 2, 0), 2, 1)
    mask = (np.arange(G)[:, None, None] < Lup[None]).astype(np.float32)[None]
    return code, mask, L.astype(np.float32)[None, None]


# ---- synthetic, seeded parameters of the two entropy models (no oracle involved: bench.py uses these too)
def make_main_params(seed, G):
    """12 layers x 3 stacked nets [weight, sigma, mu] (lic360_demo.py:104-112,302)."""
    rng = np.random.default_rng(seed)
    shapes = [(G * 1, G * 4, 5, True)] + [(G * 4, G * 4, 6, True)] * 10 + [(G * 4, G * 3, 6, False)]
    layers = []
    for C, nout, constrain, act in shapes:
        w, b, a = conv_params(rng, 3, nout, C, act=act)
        layers.append(dict(w=w, b=b, a=a, constrain=constrain))
    layers[-1]["b"][1] += 2.0          # sigma net: last bias +2 (test/model_zoo.py:263)
    return layers


def make_imp_params(seed, cpg=144, nsym=49):
    rng = np.random.default_rng(seed)
    shapes = [(1, cpg, 5, True)] + [(cpg, cpg, 6, True)] * 10 + [(cpg, nsym, 6, False)]
    layers = []
    for C, nout, constrain, act in shapes:
        w, b, a = conv_params(rng, None, nout, C, act=act)
        layers.append(dict(w=w, b=b, a=a, constrain=constrain))
    return layers
