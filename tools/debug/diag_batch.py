"""which image / group / position of a big batch decodes wrongly (diagnostic for the list-mode decode)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from util import make_latent, make_main_params
from lic360_fused import FusedCodec
G, H, W = 48, 64, 128
B, seed0 = int(os.environ.get("PB", 96)), int(os.environ.get("SEED0", 192))
layers = make_main_params(1003, G)
fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
items = [make_latent("smooth", np.random.default_rng(seed0 + i), G, H, W) for i in range(B)]
code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
for rep in range(2):
    fc.encode_async(code, mask); torch.cuda.synchronize()
    out = fc.decode_async(mask, B); torch.cuda.synchronize()
    bad = (out != code * mask)
    print("rep", rep, "err", fc.err[:B].abs().sum().item(), "bad cells", int(bad.sum().item()))
    if bad.any():
        idx = bad.nonzero()
        imgs = sorted(set(idx[:, 0].tolist()))
        print(" bad images", imgs[:20])
        b0 = imgs[0]
        sub = idx[idx[:, 0] == b0]
        planes = (sub[:, 1] + sub[:, 2] + sub[:, 3])
        first = sub[planes.argmin()]
        print(" first bad cell of image", b0, "g,y,x =", first[1:].tolist(), "plane", int(planes.min()))
os.environ["LIC360_NOSKIP"] = "1"
ref = FusedCodec(G, H, W, max_batch=B); ref.load_layers(layers)
ref.encode_async(code, mask); torch.cuda.synchronize()
print("encode bytes equal:", bool(torch.equal(ref.nbytes[:B], fc.nbytes[:B])), bool(torch.equal(ref.bytes[:B], fc.bytes[:B])))
