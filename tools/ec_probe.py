"""one encode of PB images (default 32) through the fused codec: the subject of the rocprofv3 runs behind profiles/*ec*"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
from util import make_latent, make_main_params
from lic360_fused import FusedCodec
G, H, W, B = 48, 64, 128, int(os.environ.get("PB", 32))
layers = make_main_params(1003, G)
fc = FusedCodec(G, H, W, max_batch=B); fc.load_layers(layers)
items = [make_latent(os.environ.get("MASKS", "smooth"), np.random.default_rng(i), G, H, W) for i in range(B)]   # MASKS=iid: the masks of rounds 1-5
code = torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask = torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
fc.encode_async(code, mask); torch.cuda.synchronize()
fc.profile(True)
fc.encode_async(code, mask); torch.cuda.synchronize()
for k, (ms, n) in fc.profile_read().items():
    if n: print("%-12s %4d launches  %.4f ms each" % (k, n, ms / n))
# diagnostic build (-DC16_STAMP, tools/ec_stamp.sh): cycles per phase and wave of the hidden-layer kernel, summed over one encode's 10 launches
import ctypes as C
import lic360 as lic
try:
    fn = lic._lib.lic360_c16_stamps
except AttributeError:
    fn = None
if fn is not None:
    fn.argtypes = [C.c_void_p, C.c_int]
    fn(None, 1)
    fc.encode_async(code, mask); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 8 * 10))()
    fn(buf, 0)
    st = np.array(buf, dtype=np.float64).reshape(256, 8, 10) / 10.0      # per hidden-layer launch
    names = ["step top", "ranges", "tree", "dma wait", "barrier", "post", "steps", "tiles", "preamble", "total"]
    m = st.mean((0, 1))
    print("cycles per wave and launch (mean over 256 workgroups x 8 waves):")
    for i, nme in enumerate(names): print("  %-9s %12.0f  %5.1f %%" % (nme, m[i], 100 * m[i] / m[9] if i not in (6, 7) else 0))
    print("  per step: ranges %.0f  top %.0f  wait %.0f  barrier %.0f  post %.0f ; per tile: tree %.0f  steps/tile %.2f" % (
        m[1] / m[6], m[0] / m[6], m[3] / m[6], m[4] / m[6], m[5] / m[6], m[2] / m[7], m[6] / m[7]))
    print("  total per workgroup: min %.0f  mean %.0f  max %.0f" % (st[:, :, 9].mean(1).min(), st[:, :, 9].mean(), st[:, :, 9].mean(1).max()))
    for wv in range(8): print("  wave %d (ps %d, class %d): " % (wv, wv >> 2, wv & 3) + " ".join("%10.0f" % v for v in st[:, wv, :].mean(0)))
