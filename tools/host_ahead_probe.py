"""Does the host run ahead of the GPU in the fused decode loop?  Time until the decode call returns (all launches enqueued) against the time until the
GPU is done, for 1 / 8 / 48 images (round 5: 10-11 ms against 89 / 114 / 317 ms -- the launch path is not the limiter).  Run on the GPU box."""
import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
from util import latent, make_main_params
from lic360_fused import FusedCodec
G, H, W = 48, 64, 128
for B in (1, 8, 48):
    layers = make_main_params(1003, G)
    fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
    items = [latent(np.random.default_rng(i), G, H, W) for i in range(B)]
    code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
    fc.encode_async(code, mask); torch.cuda.synchronize()
    fc.decode_async(mask, B); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); fc.decode_async(mask, B); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("B %2d decode: host call returns after %.1f ms, GPU done after %.1f ms" % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    t0 = time.perf_counter(); fc.encode_async(code, mask); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B %2d encode: host call returns after %.1f ms, GPU done after %.1f ms" % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
