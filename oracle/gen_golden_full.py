#!/usr/bin/env python3
"""Generate tests/golden/full_*.npz: bitstreams of BASELINE.json's full-size configurations produced by the CPU ORACLE
(tests/ref_codec.py over oracle/liblic360_oracle.so), so that the `-m gpu` tests can check the HIP path at those sizes
without spending minutes of oracle time on the GPU box.

  cfg2  one 512x1024 ERP latent (48x64x128),  model-idx 0          (weights seed    0)   configs[1]: encode
  cfg3  one 512x1024 ERP latent,              model-idx 3 --ssim   (weights seed 1003)   configs[2]: decode
  cfg5  one 1024x2048 ERP latent (48x128x256), model-idx 7 --ssim  (weights seed 1007)   configs[4]
  cfg2b / cfg3b  a second image of cfg2 / cfg3 at a dense (85 %) / sparse (15 %) mask
  cfg2s / cfg3s / cfg5s  the same configurations on SURVEY.md 8d's smooth importance maps (round 6)

Per case the file holds DATA only: the latent seed, the oracle's latent bitstream and importance-map bitstream (raw
bytes) and their SHA-256.  Weights and latents are regenerated from the seeds by tests/util.py (numpy Generator streams
are stable across numpy versions for default_rng/PCG64 + the distributions used there).

Run here (8 cores): cfg2/cfg3 ~1.5 min each, cfg5 ~6 min.  `python oracle/gen_golden_full.py [cfg2 cfg3 cfg5]`
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))

import ref_codec as rc  # noqa: E402
from util import make_latent, make_main_params, make_imp_params  # noqa: E402

G = 48
CASES = {
    "cfg2": dict(H=64, W=128, model_idx=0, ssim=0, latent_seed=2000),
    "cfg3": dict(H=64, W=128, model_idx=3, ssim=1, latent_seed=3000),
    "cfg5": dict(H=128, W=256, model_idx=7, ssim=1, latent_seed=5000),
    # a second pinned image per 512x1024 configuration at another mask density: dense (85 % of the groups coded)
    # and sparse (15 %) importance maps instead of the default 50 %
    "cfg2b": dict(H=64, W=128, model_idx=0, ssim=0, latent_seed=2001, mean=0.85, spread=0.10),
    "cfg3b": dict(H=64, W=128, model_idx=3, ssim=1, latent_seed=3001, mean=0.15, spread=0.10),
    # round 6: the same three configurations on SURVEY.md 8d's importance maps (smooth noise x cos(latitude), tests/util.py:latent_smooth) -- the
    # workload the bench times since round 6; the five cases above (every map cell drawn independently) stay as the adversarial family
    "cfg2s": dict(H=64, W=128, model_idx=0, ssim=0, latent_seed=2000, kind="smooth"),
    "cfg3s": dict(H=64, W=128, model_idx=3, ssim=1, latent_seed=3000, kind="smooth"),
    "cfg5s": dict(H=128, W=256, model_idx=7, ssim=1, latent_seed=5000, kind="smooth"),
}


def main():
    names = sys.argv[1:] or list(CASES)
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name in names:
        c = CASES[name]
        wseed = 1000 * c["ssim"] + c["model_idx"]
        layers = make_main_params(wseed, G)
        imp_layers = make_imp_params(wseed)
        code, mask, levels = make_latent(c.get("kind", "iid"), np.random.default_rng(c["latent_seed"]), G, c["H"], c["W"], c.get("mean", 0.5), c.get("spread", 0.25))
        t0 = time.time()
        data = rc.encode_main(code, mask, layers, G)
        t1 = time.time()
        imp = rc.encode_imp(levels, imp_layers)
        t2 = time.time()
        print("%s: latent %d bytes (%.1f s), importance %d bytes (%.1f s)" % (name, len(data), t1 - t0, len(imp), t2 - t1), flush=True)
        np.savez(os.path.join(out_dir, "full_%s.npz" % name),
                 H=c["H"], W=c["W"], G=G, weight_seed=wseed, latent_seed=c["latent_seed"], mean=c.get("mean", 0.5), spread=c.get("spread", 0.25),
                 kind=c.get("kind", "iid"),
                 bytes=np.frombuffer(data, np.uint8), sha256=hashlib.sha256(data).hexdigest(),
                 imp_bytes=np.frombuffer(imp, np.uint8), imp_sha256=hashlib.sha256(imp).hexdigest(),
                 code_sha256=hashlib.sha256(np.ascontiguousarray(code).tobytes()).hexdigest(),
                 mask_sha256=hashlib.sha256(np.ascontiguousarray(mask).tobytes()).hexdigest(),
                 oracle_encode_seconds=t1 - t0)


if __name__ == "__main__":
    main()
