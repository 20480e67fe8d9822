"""Body of __graft_entry__.smoke(): one small encode -> decode of a synthetic latent batch through the
device-resident HIP codec on cuda:0; bitstreams and symbols are checked against the CPU oracle pipeline."""
import numpy as np
import torch

import ref_codec as rc
from util import latent


def run():
    from lic360_fused import FusedCodec
    G, H, W, B = 8, 8, 16, 2
    rng = np.random.default_rng(0)
    layers = rc.make_main_params(1003, G)
    items = [latent(rng, G, H, W) for _ in range(B)]
    code = np.concatenate([it[0] for it in items], 0)
    mask = np.concatenate([it[1] for it in items], 0)
    fc = FusedCodec(G, H, W, max_batch=B)
    fc.load_layers(layers)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    ref = [rc.encode_main(code[i:i + 1], mask[i:i + 1], layers, G) for i in range(B)]
    for coder in ("device", "host"):                                    # the serial coder on the GPU (one wave per image) and on host threads (small calls)
        fc.set_coder(coder)
        streams = fc.encode(d(code), d(mask))
        for i in range(B):
            assert streams[i] == ref[i], "bitstream %d differs from the oracle (coder on the %s)" % (i, coder)
        out = fc.decode(streams, d(mask)).cpu().numpy()
        assert np.array_equal(out, code * mask), "decoded symbols differ (coder on the %s)" % coder
