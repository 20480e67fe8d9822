"""How evenly the dead-cone task lists load the eight XCDs of a decode-order launch (each XCD owns the samples n = xcd mod 8): per (layer, plane) the
wave-steps (records x chain length of their group block) of every XCD's list; a launch lasts as long as its most loaded XCD.  PB images, MASKS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from util import make_latent, make_main_params
from lic360_fused import FusedCodec
G, H, W, B = 48, 64, 128, int(os.environ.get("PB", 48))
fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(make_main_params(1003, G))
items = [make_latent(os.environ.get("MASKS", "smooth"), np.random.default_rng(i), G, H, W) for i in range(B)]
code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
fc.encode_async(code, mask); fc.decode_async(mask, B); torch.cuda.synchronize()
P = H + W + G - 2
cnt, cap = fc.debug_lists(3)
rec, _ = fc.debug_lists(4)
cnt, rec = cnt.reshape(12, P, 8), rec.reshape(12, P, 8, cap, 4)
tot_mean = tot_max = 0.0
full = 0.0
for l in range(1, 12):
    lm = lx = 0.0
    for p in range(P):
        load = np.zeros(8)
        for x in range(8):
            r = rec[l, p, x, :cnt[l, p, x], 0]
            g0 = (r & 127).astype(np.int64)
            load[x] = np.minimum(G, g0 + 7).sum()
        lm += load.mean(); lx += load.max()
    print("layer %2d: mean wave-steps per XCD and decode %9.0f, sum of per-launch maxima %9.0f  (+%.1f %%)" % (l, lm, lx, 100 * (lx / lm - 1)))
    tot_mean += lm; tot_max += lx
print("all: +%.1f %% (a launch waits for its most loaded XCD)" % (100 * (tot_max / tot_mean - 1)))
