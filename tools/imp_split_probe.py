"""step time of 144 images (3 latent sub-batches on 3 streams) with the importance maps coded (a) per sub-batch on streams of their own,
(b) as ONE batch of 144 maps on a fourth stream, (c) not at all"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
from util import latent, make_main_params, make_imp_params
from lic360_fused import FusedCodec, FusedImpCodec
G, H, W, B, NS = 48, 64, 128, 144, 3
dev = torch.device("cuda", 0)
layers, imp_layers = make_main_params(1003, G), make_imp_params(1003)
items = [latent(np.random.default_rng(i), G, H, W) for i in range(B)]
code = torch.from_numpy(np.concatenate([i[0] for i in items])).to(dev)
mask = torch.from_numpy(np.concatenate([i[1] for i in items])).to(dev)
lev = torch.from_numpy(np.concatenate([i[2] for i in items])).to(dev)
sz = B // NS
codecs = [FusedCodec(G, H, W, max_batch=sz) for _ in range(NS)]
for c in codecs: c.load_layers(layers)
ics = [FusedImpCodec(H // 2, W // 2, max_batch=sz) for _ in range(NS)]
for c in ics: c.load_layers(imp_layers)
big = FusedImpCodec(H // 2, W // 2, max_batch=B); big.load_layers(imp_layers)
st = [torch.cuda.Stream(device=dev) for _ in range(NS)]
ist = [torch.cuda.Stream(device=dev) for _ in range(NS + 1)]
def step(mode):
    for ph in (0, 1):
        if mode == "one":
            with torch.cuda.stream(ist[NS]):
                big.encode_async(lev) if ph == 0 else big.decode_async(B)
        for i in range(NS):
            sl = slice(i * sz, (i + 1) * sz)
            if mode == "split":
                with torch.cuda.stream(ist[i]):
                    ics[i].encode_async(lev[sl]) if ph == 0 else ics[i].decode_async(sz)
            with torch.cuda.stream(st[i]):
                codecs[i].encode_async(code[sl], mask[sl]) if ph == 0 else codecs[i].decode_async(mask[sl], sz)
for mode in ("split", "one", "none", "split", "one"):
    step(mode); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2): step(mode)
    torch.cuda.synchronize()
    print(mode, "%.1f ms per step" % ((time.time() - t0) / 2 * 1e3), flush=True)
print("exact", bool(torch.equal(big.levels_out[:B], lev)))
