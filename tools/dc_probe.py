"""one encode + one decode of PB images (default 32) through the fused codec: per-class launch times of the decode order kernels
(the subject of the rocprofv3 runs behind profiles/*dc*)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
from util import make_latent, make_main_params
from lic360_fused import FusedCodec
G, H, W, B = 48, int(os.environ.get("XH", 64)), int(os.environ.get("XW", 128)), int(os.environ.get("PB", 32))
layers = make_main_params(1003, G)
fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
items = [make_latent(os.environ.get("MASKS", "smooth"), np.random.default_rng(i), G, H, W) for i in range(B)]   # MASKS=iid: the masks of rounds 1-5
code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
fc.encode_async(code, mask); torch.cuda.synchronize()
fc.decode_async(mask, B); torch.cuda.synchronize()
fc.profile(True)
t0 = time.time(); fc.decode_async(mask, B); torch.cuda.synchronize(); dt = time.time() - t0
for k, (ms, n) in fc.profile_read().items():
    if n: print("%-12s %5d launches  %8.2f us each  %8.2f ms total" % (k, n, 1e3 * ms / n, ms))
print("B", B, "decode wall %.1f ms, exact %s" % (dt * 1e3, bool(torch.equal(fc.code_out[:B], code * mask))))
