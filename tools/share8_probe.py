"""enc + dec wall time of a SHORT image list (BASELINE configs[3]'s per-GPU share at 8 GPUs: 8 images) split over HIP streams in different ways, both
bitstreams, latent decode gated behind the map decode as bench.py runs it.  SPLIT="8" | "4,4" | "3,3,2"; CODER=auto|device|host"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from util import make_latent, make_main_params, make_imp_params
from lic360_fused import FusedCodec, FusedImpCodec
G, H, W = 48, 64, 128
dev = torch.device("cuda", 0)
layers, il = make_main_params(1003, G), make_imp_params(1003)
for split in os.environ.get("SPLITS", "8;4,4;3,3,2;2,2,2,2").split(";"):
    sizes = [int(v) for v in split.split(",")]
    cs, ics, cd, mk, lv, st, ist, mb = [], [], [], [], [], [], [], []
    o = 0
    for sz in sizes:
        c = FusedCodec(G, H, W, max_batch=sz); c.load_layers(layers); c.set_coder(os.environ.get("CODER", "auto")); cs.append(c)
        ic = FusedImpCodec(H // 2, W // 2, max_batch=sz); ic.load_layers(il); ics.append(ic)
        items = [make_latent("smooth", np.random.default_rng(640000 + o + i), G, H, W) for i in range(sz)]
        o += sz
        cd.append(torch.from_numpy(np.concatenate([i[0] for i in items])).to(dev)); mk.append(torch.from_numpy(np.concatenate([i[1] for i in items])).to(dev))
        lv.append(torch.from_numpy(np.concatenate([i[2] for i in items])).to(dev))
        st.append(torch.cuda.Stream(device=dev)); ist.append(torch.cuda.Stream(device=dev)); mb.append(torch.zeros((sz, G, H, W), device=dev))
    def step():
        for s in range(len(sizes)):
            with torch.cuda.stream(ist[s]):
                ics[s].encode_async(lv[s])
            with torch.cuda.stream(st[s]):
                cs[s].encode_async(cd[s], mk[s])
        for s in range(len(sizes)):
            ist[s].wait_stream(st[s])
            with torch.cuda.stream(ist[s]):
                gate = ics[s].decode_masked_async(sizes[s], mb[s])
            with torch.cuda.stream(st[s]):
                cs[s].decode_async(mb[s], sizes[s], gate=gate)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    ok = all(bool(torch.equal(cs[s].code_out[:sizes[s]], cd[s] * mk[s])) for s in range(len(sizes)))
    print("split %-10s coder %s: %.1f ms per enc+dec of %d images, exact %s" % (split, os.environ.get("CODER", "auto"), ms, sum(sizes), ok), flush=True)
    del cs, ics
    torch.cuda.empty_cache()
