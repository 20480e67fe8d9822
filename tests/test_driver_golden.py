"""D1-D4 pinned to the REFERENCE's own Python: tests/golden/driver_*.npz were written by oracle/gen_golden_drivers.py, which ran
the reference's EntEncoderFast / EntDecoder / ImpEntEncoderFast / ImpEntDecoder (test/lic360_demo.py:95-290) with weights pushed
through its cast_entropy_parameter / cast_imp_entropy_parameter (:296-322) on the CPU (the extension's kernels replaced by the
oracle's, the reference's arithmetic coder cross-checking every bitstream).  What must reproduce those files:
  * CPU: tests/ref_codec.py -- the restatement of the drivers the other parity tests compare the HIP path with;
  * -m gpu: the op-level drivers of lic360_codec.py (same classes on the C ABI) and the device-resident FusedCodec / FusedImpCodec.
Symbol order, the (data - 3.5) * mask input, the [weight, sigma, mu] batch order, file_value 3.5, Scale(-1, 2/47), the
floor((b + 1) / scale + 1e-5) -> Imp2mask -> Dtow tail and the checkpoint key mapping are thereby the reference's."""
import glob
import os

import numpy as np
import pytest

import ref_codec as rc

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "driver_*.npz")))


def load(path):
    g = np.load(path)
    G, H, W = int(g["G"]), int(g["H"]), int(g["W"])
    return g, G, H, W, rc.make_main_params(int(g["wseed"]), G), rc.make_imp_params(int(g["wseed"]))


def test_golden_files_exist():
    assert len(GOLD) >= 3


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_ref_codec_reproduces_the_reference_drivers(path):
    g, G, H, W, layers, imp_layers = load(path)
    code, mask, levels = g["code"], g["mask"], g["levels"]
    assert rc.encode_main(code, mask, layers, G) == g["latent_bytes"].tobytes()
    assert rc.encode_imp(levels, imp_layers) == g["imp_bytes"].tobytes()
    lv = rc.decode_imp(g["imp_bytes"].tobytes(), imp_layers, H // 2, W // 2)
    assert np.array_equal(lv, levels)
    Lup = np.repeat(np.repeat(lv[0, 0], 2, 0), 2, 1)                       # Dtow(2)(Imp2mask(levels)): mask[g, y, x] = g < L[y/2, x/2]
    assert np.array_equal((np.arange(G)[:, None, None] < Lup[None]).astype(np.float32)[None], g["decoded_mask"])
    assert np.array_equal(rc.decode_main(g["latent_bytes"].tobytes(), g["decoded_mask"], layers, G), g["decoded_code"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_hip_drivers_reproduce_the_reference_drivers(tmp_path, path):
    import torch
    import lic360_codec as lc
    from lic360_fused import FusedCodec, FusedImpCodec
    g, G, H, W, layers, imp_layers = load(path)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    code, mask, levels = g["code"], g["mask"], g["levels"]
    lat, imp = g["latent_bytes"].tobytes(), g["imp_bytes"].tobytes()
    # ---- op-level drivers (the reference's structure on the C ABI), parameters through OUR cast_* from a training-layout checkpoint
    ck = {}
    for b, pre in enumerate(("ent.weight_net", "ent.delta_net", "ent.mean_net")):
        for k, src in lc._key_map(pre).items():
            layer = 0 if k.startswith("net.0.") else (11 if k.startswith("net.6.") else 2 * int(k.split(".")[1]) - (1 if ".conv1." in k else 0))
            part = {"weight": "w", "bias": "b", "relu": "a"}[k.split(".")[-1]]
            ck[src] = dev(layers[layer][part][b])
    enc = lc.EntEncoderFast(G, 8, 0)
    enc.load_state_dict(lc.cast_entropy_parameter(ck, enc.state_dict()))
    f, fi = str(tmp_path / "lat"), str(tmp_path / "lat_imp")
    enc.start(f)
    enc(dev(code), dev(mask))
    assert open(f, "rb").read() == lat
    ienc = lc.ImpEntEncoderFast(48, 0)
    rc.load_into_driver(ienc, imp_layers)
    ienc.start(fi)
    ienc(dev(levels))
    assert open(fi, "rb").read() == imp
    idec = lc.ImpEntDecoder(48, 0)
    rc.load_into_driver(idec, imp_layers)
    idec.start(fi)
    tmask = idec(H // 2, W // 2)
    assert np.array_equal(tmask.cpu().numpy(), g["decoded_mask"])
    dec = lc.EntDecoder(G, 8, 0)
    dec.load_state_dict(lc.cast_entropy_parameter(ck, dec.state_dict()))
    dec.start(f)
    assert np.array_equal(dec(tmask).cpu().numpy(), g["decoded_code"])
    # ---- device-resident codecs
    fc = FusedCodec(G, H, W, max_batch=4)
    fc.load_layers(layers)
    assert fc.encode(dev(code), dev(mask))[0] == lat
    assert np.array_equal(fc.decode([lat], dev(g["decoded_mask"])).cpu().numpy(), g["decoded_code"])
    fic = FusedImpCodec(H // 2, W // 2, max_batch=4)
    fic.load_layers(imp_layers)
    assert fic.encode(dev(levels))[0] == imp
    assert np.array_equal(fic.decode([imp]).cpu().numpy(), levels)
    if H >= 28:
        # the 64-row golden as a batch of 16 copies: more than 128 three-group tasks per launch, so the decode kernel leaves its latency mode and
        # runs its throughput schedule -- full-lane diagonals, the samples of an XCD's list packed end to end over the lanes (8 | samples per net; 16 images: the dead-cone task records)
        fcb = FusedCodec(G, H, W, max_batch=16)
        fcb.load_layers(layers)
        rep = lambda a: np.ascontiguousarray(np.repeat(a, 16, 0))
        assert fcb.encode(dev(rep(code)), dev(rep(mask))) == [lat] * 16
        assert np.array_equal(fcb.decode([lat] * 16, dev(rep(g["decoded_mask"]))).cpu().numpy(), rep(g["decoded_code"]))
