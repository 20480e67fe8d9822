"""Randomised soak of the dead-cone skip (csrc/need.h): random latent shapes, batch sizes and mask families; the fused codec with the skip on (activation
buffers poisoned first) must produce the bytes of a codec with the skip off and decode them exactly, in list mode (>= 16 images, 8 | batch) and below it.
usage: N=120 SEED=1 python tools/list_soak.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from util import latent, latent_smooth, make_main_params
from lic360_fused import FusedCodec

N, SEED = int(os.environ.get("N", 120)), int(os.environ.get("SEED", 1))
rng = np.random.default_rng(SEED)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
fails = 0
t0 = time.time()
for case in range(N):
    G = int(rng.choice([5, 6, 9, 12, 24, 48]))
    H = int(rng.choice([4, 8, 14, 20, 32, 50, 64]))
    W = int(rng.choice([6, 10, 16, 24, 40, 64, 128])) if G < 48 else int(rng.choice([8, 16, 24]))
    B = int(rng.choice([1, 3, 8, 16, 24, 32, 40, 64])) if G * H * W < 40000 else int(rng.choice([2, 16, 24]))
    fam = str(rng.choice(["smooth", "iid", "sparse", "dense", "blob", "empty1"]))
    cs, ms = [], []
    for i in range(B):
        r = np.random.default_rng(int(rng.integers(1 << 30)))
        if fam == "smooth" and H % 2 == 0 and W % 2 == 0:
            c, m, _ = latent_smooth(r, G, H, W)
        elif fam == "sparse":
            c, m, _ = latent(r, G, H - H % 2, W - W % 2, 0.1, 0.1) if H % 2 == 0 and W % 2 == 0 else latent(r, G, H, W, 0.1, 0.1) if False else (None, None, None)
        elif fam == "dense":
            c, m, _ = latent(r, G, H, W, 0.9, 0.1) if H % 2 == 0 and W % 2 == 0 else (None, None, None)
        else:
            c = m = None
        if c is None:                                                  # any geometry: a prefix mask from a random level map (blob: one live patch)
            c = np.clip(np.rint(r.normal(3.5, 1.2, (1, G, H, W))), 0, 7).astype(np.float32)
            if fam == "blob":
                L = np.zeros((H, W), np.int64)
                y0, x0 = int(r.integers(H)), int(r.integers(W))
                L[y0:y0 + max(1, H // 3), x0:x0 + max(1, W // 3)] = int(r.integers(1, G + 1))
            else:
                L = np.clip(np.rint(G * 0.5 + G * 0.3 * r.standard_normal((H, W))), 0, G).astype(np.int64)
            m = (np.arange(G)[:, None, None] < L[None]).astype(np.float32)[None]
        if fam == "empty1" and i == B // 2:
            m = np.zeros_like(m)
        cs.append(c); ms.append(m)
    code, mask = np.concatenate(cs, 0), np.concatenate(ms, 0)
    layers = make_main_params(int(rng.integers(1 << 20)), G)
    os.environ["LIC360_NOSKIP"] = "1"
    ref = FusedCodec(G, H, W, max_batch=B); ref.load_layers(layers)
    del os.environ["LIC360_NOSKIP"]
    fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
    want = ref.encode(dev(code), dev(mask))
    fc.debug_fill(1e10 if case % 2 else -1e10)
    got = fc.encode(dev(code), dev(mask))
    fc.debug_fill(-1e10 if case % 2 else 1e10)
    out = fc.decode(want, dev(mask)).cpu().numpy()
    ok = got == want and np.array_equal(out, code * mask)
    if not ok:
        fails += 1
        print("FAIL case %d: G %d H %d W %d B %d %s skip_active %d bytes_equal %s decode_exact %s" % (case, G, H, W, B, fam, fc.skip_active(), got == want, np.array_equal(out, code * mask)), flush=True)
    del ref, fc
    if case % 20 == 19:
        print("... %d cases, %d failures, %.0f s" % (case + 1, fails, time.time() - t0), flush=True)
print("list soak: %d cases, %d failures" % (N, fails))
sys.exit(1 if fails else 0)
