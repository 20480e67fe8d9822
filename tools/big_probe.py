"""1024x2048 ERP latents (48x128x256): encode + decode round trip through the fused codec with timings (config 5 of BASELINE.json)."""
import os, sys, time
sys.path.insert(0, "360-image-compression_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch, numpy as np
import ref_codec as rc
from lic360_fused import FusedCodec
from util import make_latent
G, H, W = 48, 128, 256
B = int(os.environ.get("PB", "8"))
layers = rc.make_main_params(1007, G)
fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
items = [make_latent(os.environ.get("MASKS", "smooth"), np.random.default_rng(i), G, H, W) for i in range(B)]   # MASKS=iid: the masks of rounds 1-5
code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    fc.encode_async(code, mask); torch.cuda.synchronize(); t1 = time.time()
    fc.decode_async(mask, B); torch.cuda.synchronize(); t2 = time.time()
ok = bool(torch.equal(fc.code_out[:B], code * mask)) and int(fc.err[:B].abs().sum().item()) == 0
px = B * 1024 * 2048
print("B", B, "encode %.1f ms decode %.1f ms -> %.2f Mpixel/s (single stream), round trip %s, mean bytes %.0f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, px / (t2 - t0) / 1e6, "exact" if ok else "MISMATCH", float(fc.nbytes[:B].float().mean().item())))
