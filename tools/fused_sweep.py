"""One-off robustness sweep: fused codec vs oracle pipeline on odd shapes (run on the GPU box)."""
import sys
sys.path.insert(0, "360-image-compression_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np, torch
import ref_codec as rc
from lic360_fused import FusedCodec
from util import latent
cases = [(3, 2, 8, 1), (5, 10, 8, 2), (7, 34, 14, 3), (4, 62, 8, 2), (6, 64, 10, 1), (9, 20, 36, 2), (48, 4, 6, 1), (13, 12, 66, 2), (2, 64, 64, 1), (10, 30, 8, 5),
         # batches that engage the decode kernel's tape packing (8 | images; tapes of 5, 4, 3, 2, 6 samples; short, cut and 64-row windows)
         (12, 50, 12, 40), (12, 34, 10, 32), (9, 64, 30, 24), (20, 46, 62, 16), (15, 64, 20, 48), (6, 28, 40, 56)]
if len(sys.argv) > 1:
    cases = cases[int(sys.argv[1]):]
bad = 0
for ci, (G, H, W, B) in enumerate(cases):
    rng = np.random.default_rng(100 + ci)
    layers = rc.make_main_params(3000 + ci, G)
    items = [latent(rng, G, H, W) for _ in range(B)]
    code = np.concatenate([it[0] for it in items], 0); mask = np.concatenate([it[1] for it in items], 0)
    fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
    streams = fc.encode(torch.from_numpy(code).cuda(), torch.from_numpy(mask).cuda())
    check = range(B) if B <= 8 else sorted({0, 1, B // 2, B - 2, B - 1})          # the oracle encodes one image at a time: sample the big batches
    ok = all(streams[i] == rc.encode_main(code[i:i + 1], mask[i:i + 1], layers, G) for i in check)
    out = fc.decode(streams, torch.from_numpy(mask).cuda()).cpu().numpy()
    ok2 = np.array_equal(out, code * mask)
    print((G, H, W, B), "bitstreams", "OK" if ok else "DIFF", "decode", "OK" if ok2 else "DIFF", flush=True)
    bad += (not ok) + (not ok2)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
