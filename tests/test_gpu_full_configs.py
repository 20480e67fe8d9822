"""BASELINE.json's configurations at their FULL sizes on the HIP path, against bitstreams the CPU oracle produced in the
build container (tests/golden/full_*.npz, oracle/gen_golden_full.py; 27 s / 109 s of oracle time per image there):

  cfg2  configs[1]  single 512x1024 ERP encode,  model-idx 0          -> latent + importance bitstreams == oracle bytes
  cfg3  configs[2]  single 512x1024 ERP decode,  model-idx 3 --ssim   -> decode of the ORACLE's bytes == the symbols
  cfg5  configs[4]  1024x2048 ERP (48x128x256 latent), model-idx 7 --ssim -> both directions
  cfg4  configs[3]  is the batch form of cfg3: test_batch_of_cfg3_images checks 8 images per call (one GPU's share)
  cfg2b / cfg3b  a second oracle-pinned image of cfg2 / cfg3 at a dense / sparse importance mask (both directions for cfg3b via the batch test)
  cfg2s / cfg3s / cfg5s  the same configurations on SURVEY.md 8d's smooth importance maps (round 6: the masks the bench times; the i.i.d. cases
                 above stay as the adversarial family -- the dead-cone skip finds next to nothing in them)
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from util import latent, make_latent, make_main_params, make_imp_params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def load(name):
    g = np.load(os.path.join(GOLD, "full_%s.npz" % name))
    G, H, W = int(g["G"]), int(g["H"]), int(g["W"])
    dens = (float(g["mean"]), float(g["spread"])) if "mean" in g.files else (0.5, 0.25)
    kind = str(g["kind"]) if "kind" in g.files else "iid"
    code, mask, levels = make_latent(kind, np.random.default_rng(int(g["latent_seed"])), G, H, W, *dens)
    # the fixtures hold digests of the inputs they were made from: a drifting generator fails here, not as a byte mismatch
    assert hashlib.sha256(code.tobytes()).hexdigest() == str(g["code_sha256"])
    assert hashlib.sha256(mask.tobytes()).hexdigest() == str(g["mask_sha256"])
    wseed = int(g["weight_seed"])
    return g, (G, H, W), code, mask, levels, make_main_params(wseed, G), make_imp_params(wseed)


def codecs(shape, layers, imp_layers, batch=1):
    from lic360_fused import FusedCodec, FusedImpCodec
    G, H, W = shape
    fc = FusedCodec(G, H, W, max_batch=batch)
    fc.load_layers(layers)
    ic = FusedImpCodec(H // 2, W // 2, max_batch=batch)
    ic.load_layers(imp_layers)
    return fc, ic


@pytest.mark.parametrize("name", ["cfg2", "cfg2b", "cfg5", "cfg2s", "cfg5s"])
def test_full_size_encode_bytes_equal_oracle(name):
    g, shape, code, mask, levels, layers, imp_layers = load(name)
    fc, ic = codecs(shape, layers, imp_layers)
    data = fc.encode(dev(code), dev(mask))[0]
    assert len(data) == len(g["bytes"]) and hashlib.sha256(data).hexdigest() == str(g["sha256"])
    assert data == g["bytes"].tobytes()
    imp = ic.encode(dev(levels))[0]
    assert imp == g["imp_bytes"].tobytes() and hashlib.sha256(imp).hexdigest() == str(g["imp_sha256"])


@pytest.mark.parametrize("name", ["cfg3", "cfg3b", "cfg5", "cfg3s", "cfg5s"])
def test_full_size_decode_of_oracle_bytes(name):
    g, shape, code, mask, levels, layers, imp_layers = load(name)
    fc, ic = codecs(shape, layers, imp_layers)
    out = fc.decode([g["bytes"].tobytes()], dev(mask)).cpu().numpy()
    assert np.array_equal(out, code * mask)
    lv = ic.decode([g["imp_bytes"].tobytes()]).cpu().numpy()
    assert np.array_equal(lv, levels)


def test_batch_of_cfg3_images():
    """configs[3]'s per-GPU share (8 images of the 64): image 0 is the oracle-pinned cfg3 image, the others are pinned by
    the round trip and by image 0 staying byte-identical inside a batch (images are independent: SURVEY.md §8e)."""
    g, shape, code, mask, levels, layers, imp_layers = load("cfg3")
    G, H, W = shape
    gb, _, code_b, mask_b, _, _, _ = load("cfg3b")                            # image 1: the second oracle-pinned image (sparse mask, same weights)
    codes, masks = [code, code_b], [mask, mask_b]
    for i in range(2, 8):
        c, m, _ = latent(np.random.default_rng(3000 + i), G, H, W)
        codes.append(c)
        masks.append(m)
    code8, mask8 = np.concatenate(codes, 0), np.concatenate(masks, 0)
    fc, _ = codecs(shape, layers, imp_layers, batch=8)
    streams = fc.encode(dev(code8), dev(mask8))
    assert streams[0] == g["bytes"].tobytes() and streams[1] == gb["bytes"].tobytes()
    assert np.array_equal(fc.decode(streams, dev(mask8)).cpu().numpy(), code8 * mask8)


def test_container_round_trip_of_the_fused_codecs_bitstreams(tmp_path):
    """row f2: the two bitstreams the FusedCodec / FusedImpCodec produce for cfg2s, packed into the single-file container, written,
    read back, unpacked -- the payloads are the oracle's bytes (what the reference would have written to `<code>` and `<code>_imp`,
    test/lic360_demo.py:361-365) -- and decoded on the GPU from the unpacked payloads: map first, the latent under the mask the decoded map gives."""
    import lic360_container as box
    g, shape, code, mask, levels, layers, imp_layers = load("cfg2s")
    G, H, W = shape
    fc, ic = codecs(shape, layers, imp_layers)
    lat, imp = fc.encode(dev(code), dev(mask))[0], ic.encode(dev(levels))[0]
    path = str(tmp_path / "erp.l360")
    box.write_file(path, lat, imp, 8 * H, 8 * W, model_idx=0, ssim=False)
    d = box.read_file(path)
    assert (d["height"], d["width"], d["model_idx"], d["ssim"]) == (512, 1024, 0, False)
    assert d["latent"] == g["bytes"].tobytes() and d["imp"] == g["imp_bytes"].tobytes()
    # the reference's two-file form and back
    box.to_reference_files(open(path, "rb").read(), str(tmp_path / "code"))
    assert open(str(tmp_path / "code"), "rb").read() == g["bytes"].tobytes() and open(str(tmp_path / "code_imp"), "rb").read() == g["imp_bytes"].tobytes()
    lv = ic.decode([d["imp"]])
    assert np.array_equal(lv.cpu().numpy(), levels)
    import lic360
    m2 = lic360.DtowOp(2, True, 0, False).forward(lic360.Imp2maskOp(48, 192, 0, False).forward(lv)[0])[0]     # lic360_demo.py:283-287
    assert np.array_equal(m2.cpu().numpy(), mask)
    out = fc.decode([d["latent"]], m2.contiguous()).cpu().numpy()
    assert np.array_equal(out, code * mask)
    bad = bytearray(open(path, "rb").read())
    bad[40] ^= 0x10
    with pytest.raises(box.ContainerError):
        box.unpack(bytes(bad))
