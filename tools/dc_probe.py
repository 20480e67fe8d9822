import os, sys
sys.path.insert(0, "360-image-compression_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch, numpy as np
import ref_codec as rc
from lic360_fused import FusedCodec
from util import latent
G,H,W = 48,64,128
B = int(os.environ.get("PB", "32"))
layers = rc.make_main_params(1003, G)
fc = FusedCodec(G,H,W,max_batch=B); fc.load_layers(layers)
items=[latent(np.random.default_rng(i),G,H,W) for i in range(B)]
code=torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask=torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
fc.encode_async(code,mask); torch.cuda.synchronize()
fc.profile(True)
import time
t0=time.time(); fc.decode_async(mask,B); torch.cuda.synchronize(); dt=time.time()-t0
p=fc.profile_read()
print("B", B, "dc hidden: %.1f us per launch, total %.1f ms; decode wall %.1f ms" % (1e3*p["dc_ms"]/p["dc_launches"], p["dc_ms"], dt*1e3))
