import os, sys, time
sys.path.insert(0, "360-image-compression_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch, numpy as np
import ref_codec as rc
from lic360_fused import FusedCodec
G,H,W,B = 48,64,128,8
layers = rc.make_main_params(1003, G)
fc = FusedCodec(G,H,W,max_batch=B); fc.load_layers(layers)
from util import latent
items=[latent(np.random.default_rng(i),G,H,W) for i in range(B)]
code=torch.from_numpy(np.concatenate([i[0] for i in items])).cuda(); mask=torch.from_numpy(np.concatenate([i[1] for i in items])).cuda()
fc.encode_async(code,mask); torch.cuda.synchronize()
fc.profile(True)
fc.encode_async(code,mask); torch.cuda.synchronize()
p=fc.profile_read()
print("ec hidden ms per launch (B=8): %.3f" % (p["ec_ms"]/p["ec_launches"]))
