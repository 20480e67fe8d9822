"""stress of the flaky wrong-decode seen with 96-image sub-batches on three streams: counts steps whose round trip is not exact, per mode
  full    latent + importance codecs on streams of their own, latent decode gated behind the map decode (bench.py's run())
  latent  latent codecs only, masks from the host-side tensors (three streams)
  same    importance codec on its sub-batch's stream
usage: PB=96 NS=3 STEPS=30 python tools/debug/stress_big.py full latent"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from util import make_latent, make_main_params, make_imp_params
from lic360_fused import FusedCodec, FusedImpCodec
G, H, W = 48, 64, 128
PB, NS, STEPS = int(os.environ.get("PB", 96)), int(os.environ.get("NS", 3)), int(os.environ.get("STEPS", 30))
dev = torch.device("cuda", 0)
layers, il = make_main_params(1003, G), make_imp_params(1003)
cs, ics, cd, mk, lv, st, ist, mb, ev = [], [], [], [], [], [], [], [], []
for s in range(NS):
    c = FusedCodec(G, H, W, max_batch=PB); c.load_layers(layers); cs.append(c)
    ic = FusedImpCodec(H // 2, W // 2, max_batch=PB); ic.load_layers(il); ics.append(ic)
    items = [make_latent("smooth", np.random.default_rng(1000 * s + i), G, H, W) for i in range(PB)]
    cd.append(torch.from_numpy(np.concatenate([i[0] for i in items])).to(dev)); mk.append(torch.from_numpy(np.concatenate([i[1] for i in items])).to(dev))
    lv.append(torch.from_numpy(np.concatenate([i[2] for i in items])).to(dev))
    st.append(torch.cuda.Stream(device=dev)); ist.append(torch.cuda.Stream(device=dev))
    mb.append([torch.zeros((PB, G, H, W), device=dev) for _ in range(2)]); ev.append([torch.cuda.Event(), torch.cuda.Event()])
torch.cuda.synchronize()
flip = 0
def step(mode):
    global flip
    flip ^= 1
    for s in range(NS):
        if mode != "latent":
            with torch.cuda.stream(ist[s] if mode == "full" else st[s]):
                ics[s].encode_async(lv[s])
        with torch.cuda.stream(st[s]):
            cs[s].encode_async(cd[s], mk[s])
    for s in range(NS):
        if mode == "latent":
            with torch.cuda.stream(st[s]):
                cs[s].decode_async(mk[s], PB)
        elif mode == "same":
            with torch.cuda.stream(st[s]):
                ics[s].decode_masked_async(PB, mb[s][flip]); cs[s].decode_async(mb[s][flip], PB)
        else:
            ist[s].wait_event(ev[s][flip])
            with torch.cuda.stream(ist[s]):
                gate = ics[s].decode_masked_async(PB, mb[s][flip])
            with torch.cuda.stream(st[s]):
                cs[s].decode_async(mb[s][flip], PB, gate=gate); ev[s][flip].record(st[s])
for mode in sys.argv[1:]:
    bad = 0
    t0 = time.time()
    for k in range(STEPS):
        step(mode)
        torch.cuda.synchronize()
        for s in range(NS):
            w = (cs[s].code_out[:PB] != cd[s] * mk[s]).flatten(1).any(1).nonzero().flatten().tolist()
            if w:
                bad += 1
                print("  mode %s step %d sub-batch %d: wrong images %s err %s" % (mode, k, s, w[:8], cs[s].err[:PB].abs().sum().item()), flush=True)
    print("mode %s: %d bad sub-batch decodes in %d steps (%.0f s)" % (mode, bad, STEPS, time.time() - t0), flush=True)
