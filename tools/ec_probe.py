"""one encode of PB images (default 32) through the fused codec: the subject of the rocprofv3 runs behind profiles/*ec*"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
from util import latent, make_main_params
from lic360_fused import FusedCodec
G, H, W, B = 48, 64, 128, int(os.environ.get("PB", 32))
layers = make_main_params(1003, G)
fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
items = [latent(np.random.default_rng(i), G, H, W) for i in range(B)]
code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
fc.encode_async(code, mask); torch.cuda.synchronize()
fc.profile(True)
fc.encode_async(code, mask); torch.cuda.synchronize()
for k, (ms, n) in fc.profile_read().items():
    if n: print("%-12s %4d launches  %.4f ms each" % (k, n, ms / n))
