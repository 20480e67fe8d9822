"""Importance-map stream of 512x1024 ERPs (32x64 maps, 144 hidden channels, 49 levels) through FusedImpCodec: timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
import ref_codec as rc
from lic360_fused import FusedImpCodec
B = int(os.environ.get("PB", "32"))
MH, MW = int(os.environ.get("MH", "32")), int(os.environ.get("MW", "64"))          # 64 x 128 = the maps of 1024x2048 ERPs
layers = rc.make_imp_params(1003)
fc = FusedImpCodec(MH, MW, max_batch=B); fc.load_layers(layers)
rng = np.random.default_rng(0)
lv = torch.from_numpy(np.clip(np.rint(24 + 12 * rng.standard_normal((B, 1, MH, MW))), 0, 48).astype(np.float32)).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    fc.encode_async(lv); torch.cuda.synchronize(); t1 = time.time()
    fc.decode_async(B); torch.cuda.synchronize(); t2 = time.time()
ok = bool(torch.equal(fc.levels_out[:B], lv)) and int(fc.err[:B].abs().sum().item()) == 0
print("B", B, "encode %.1f ms decode %.1f ms, round trip %s, mean bytes %.0f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, "exact" if ok else "MISMATCH", float(fc.nbytes[:B].float().mean().item())))
