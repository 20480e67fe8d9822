"""D1-D4 driver parity: the op-level codec drivers (lic360_codec.py, same structure as
test/lic360_demo.py) produce byte-identical bitstreams and identical decoded symbols to the oracle
pipeline on seeded synthetic latents."""
import numpy as np
import pytest
import torch

import ref_codec as rc
from util import latent

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("G,H,W,seed", [(6, 8, 12, 1), (8, 6, 10, 2)])
def test_main_latent_roundtrip(tmp_path, G, H, W, seed):
    import lic360_codec as lc
    rng = np.random.default_rng(seed)
    code, mask, _ = latent(rng, G, H, W)
    layers = rc.make_main_params(1000 + seed, G)
    ref_bytes = rc.encode_main(code, mask, layers, G)
    enc = lc.EntEncoderFast(G, 8, 0)
    rc.load_into_driver(enc, layers)
    f = str(tmp_path / "code.bin")
    enc.start(f)
    enc(dev(code), dev(mask))
    got = open(f, "rb").read()
    assert got == ref_bytes
    decd = lc.EntDecoder(G, 8, 0)
    rc.load_into_driver(decd, layers)
    decd.start(f)
    out = decd(dev(mask)).cpu().numpy()
    assert np.array_equal(out, code * mask)
    assert np.array_equal(rc.decode_main(ref_bytes, mask, layers, G), code * mask)


def test_importance_map_roundtrip(tmp_path):
    import lic360_codec as lc
    rng = np.random.default_rng(5)
    H, W = 6, 9
    levels = rng.integers(0, 13, (1, 1, H, W)).astype(np.float32)
    layers = rc.make_imp_params(77, cpg=36, nsym=13)
    ref_bytes = rc.encode_imp(levels, layers, nsym=13)
    enc = lc.ImpEntEncoderFast(12, 0)
    rc.load_into_driver(enc, layers)
    f = str(tmp_path / "imp.bin")
    enc.start(f)
    enc(dev(levels))
    assert open(f, "rb").read() == ref_bytes
    dec = lc.ImpEntDecoder(12, 0)
    rc.load_into_driver(dec, layers)
    dec.start(f)
    mask_up = dec(H, W).cpu().numpy()
    assert np.array_equal(dec.last_levels.cpu().numpy(), levels)
    assert np.array_equal(rc.decode_imp(ref_bytes, layers, H, W, nsym=13), levels)
    # mask_up[g,y,x] = g < L[y/2,x/2]   
    Lup = np.repeat(np.repeat(levels[0, 0], 2, 0), 2, 1)
    assert np.array_equal(mask_up[0], (np.arange(12)[:, None, None] < Lup[None]).astype(np.float32))


def test_cast_entropy_parameter():
    import lic360_codec as lc
    enc = lc.EntEncoderFast(4, 8, 0)
    nd = {k: v.clone() for k, v in enc.state_dict().items()}
    pd = {}
    for idx, prex in enumerate(["ent.weight_net", "ent.delta_net", "ent.mean_net"]):
        for k, src in lc._key_map(prex).items():
            pd[src] = torch.full_like(nd[k][idx], float(idx + 1))
    out = lc.cast_entropy_parameter(pd, nd)
    for k, v in out.items():
        for idx in range(3):
            assert torch.all(v[idx] == idx + 1), k
    assert set(out.keys()) == {"net.0.weight", "net.0.bias", "net.0.relu", "net.6.weight", "net.6.bias"} | {
        "net.%d.%s.%s" % (b, c, p) for b in range(1, 6) for c in ("conv1", "conv2") for p in ("weight", "bias", "relu")}


@pytest.mark.parametrize("G,H,W,B,seed", [(6, 8, 12, 3, 11), (8, 6, 10, 1, 12), (48, 8, 16, 2, 13),
                                          (4, 64, 12, 1, 16),      # full-height anti-diagonals: every lane of the wave is an image row
                                          (5, 66, 10, 2, 14),      # H > 64: two row segments per diagonal
                                          (3, 130, 8, 1, 18),      # three row segments, the last one 6 rows
                                          (4, 64, 6, 1, 15),       # W < 7: row-major encode kernel, diagonal decode kernel
                                          (3, 66, 68, 1, 17),      # H, W > 64 (the 1024x2048 regime)
                                          (4, 64, 12, 16, 19),     # 16 images: short diagonals at both image corners carry two samples per wave (tapes of 2)
                                          (6, 8, 12, 16, 20),      # every diagonal short: all decode tasks carry two samples
                                          (3, 130, 8, 16, 21),     # three row segments, the last (6 rows) with two samples per wave
                                          (4, 66, 10, 16, 22),     # two row segments, the last (4 rows) packed
                                          (6, 28, 40, 24, 23),     # 8 | images: the decode kernel tape-packs its samples (tapes of 3; round 5)
                                          (6, 50, 12, 40, 24)])    # ... tapes of 5, windows cut between tasks
@pytest.mark.parametrize("coder", ["device", "host"])
def test_fused_codec_matches_oracle(G, H, W, B, seed, coder):
    """Device-resident codec: byte-identical bitstreams to the oracle pipeline, exact decode -- with the serial coder phases on the GPU (one wave per
    image) and on host threads (the reference's own place for them, coder.cpp:70-113: records D2H / per-plane tables through pinned memory)."""
    from lic360_fused import FusedCodec
    rng = np.random.default_rng(seed)
    layers = rc.make_main_params(2000 + seed, G)
    items = [latent(rng, G, H, W) for _ in range(B)]
    code = np.concatenate([it[0] for it in items], 0)
    mask = np.concatenate([it[1] for it in items], 0)
    if B > 1:
        mask[1] = 0.0                      # one fully masked image -> bare terminator byte
    fc = FusedCodec(G, H, W, max_batch=max(4, B))
    fc.load_layers(layers)
    fc.set_coder(coder)
    streams = fc.encode(dev(code), dev(mask))
    for i in range(B):
        assert streams[i] == rc.encode_main(code[i:i + 1], mask[i:i + 1], layers, G), "image %d" % i
    if B > 1:
        assert streams[1] == b"\x80"
    out = fc.decode(streams, dev(mask)).cpu().numpy()
    assert np.array_equal(out, code * mask)


@pytest.mark.parametrize("G,H,W,B,seed", [(6, 8, 12, 2, 21), (4, 66, 10, 1, 22)])
def test_fused_codec_generic_kernel_fallback(monkeypatch, G, H, W, B, seed):
    """The fall-back path for net shapes the specialised kernels do not cover -- the generic 16x16x4 kernels of cconv_kernels.hip on plain
    NCHW / diagonal-major planes with the separate table kernel -- forced with LIC360_FUSED_CONV=16 (read once, when a codec is created):
    the same bitstreams, exact decode."""
    from lic360_fused import FusedCodec
    rng = np.random.default_rng(seed)
    layers = rc.make_main_params(2000 + seed, G)
    items = [latent(rng, G, H, W) for _ in range(B)]
    code = np.concatenate([it[0] for it in items], 0)
    mask = np.concatenate([it[1] for it in items], 0)
    ref = [rc.encode_main(code[i:i + 1], mask[i:i + 1], layers, G) for i in range(B)]
    monkeypatch.setenv("LIC360_FUSED_CONV", "16")
    fc = FusedCodec(G, H, W, max_batch=B)
    fc.load_layers(layers)
    streams = fc.encode(dev(code), dev(mask))
    assert streams == ref
    assert np.array_equal(fc.decode(streams, dev(mask)).cpu().numpy(), code * mask)


@pytest.mark.parametrize("gstep", ["1", "3"])
@pytest.mark.parametrize("G,H,W,B,seed", [(6, 8, 12, 3, 41), (48, 8, 16, 2, 42), (7, 64, 12, 1, 43), (4, 66, 10, 1, 44)])
def test_fused_codec_decode_task_granularity(monkeypatch, gstep, G, H, W, B, seed):
    """decode-order conv tasks of three groups (throughput) or of one group (latency mode, picked automatically for few samples):
    LIC360_DC_GSTEP forces either; both decode the oracle's bitstreams exactly"""
    from lic360_fused import FusedCodec
    rng = np.random.default_rng(seed)
    layers = rc.make_main_params(2000 + seed, G)
    items = [latent(rng, G, H, W) for _ in range(B)]
    code = np.concatenate([it[0] for it in items], 0)
    mask = np.concatenate([it[1] for it in items], 0)
    ref = [rc.encode_main(code[i:i + 1], mask[i:i + 1], layers, G) for i in range(B)]
    monkeypatch.setenv("LIC360_DC_GSTEP", gstep)
    fc = FusedCodec(G, H, W, max_batch=B)
    fc.load_layers(layers)
    assert np.array_equal(fc.decode(ref, dev(mask)).cpu().numpy(), code * mask)


@pytest.mark.parametrize("cpg,nsym,H,W,B,seed", [(8, 49, 8, 12, 3, 31), (144, 49, 4, 6, 1, 32), (16, 10, 32, 10, 2, 33)])
def test_fused_importance_codec_matches_oracle(cpg, nsym, H, W, B, seed):
    """Device-resident importance-map stream: byte-identical to the oracle's ImpEntEncoderFast pipeline, exact decode."""
    from lic360_fused import FusedImpCodec
    rng = np.random.default_rng(seed)
    layers = rc.make_imp_params(4000 + seed, cpg, nsym)
    levels = rng.integers(0, nsym, (B, 1, H, W)).astype(np.float32)
    fc = FusedImpCodec(H, W, max_batch=4, hidden_channels=cpg, nsym=nsym)
    fc.load_layers(layers)
    streams = fc.encode(dev(levels))
    for i in range(B):
        assert streams[i] == rc.encode_imp(levels[i:i + 1], layers, nsym), "image %d" % i
    out = fc.decode(streams).cpu().numpy()
    assert np.array_equal(out, levels)
    assert np.array_equal(rc.decode_imp(streams[0], layers, H, W, nsym), levels[0:1])


@pytest.mark.parametrize("G,MH,MW,B,seed", [(12, 8, 12, 1, 71), (12, 8, 12, 3, 72), (24, 6, 10, 2, 73)])
def test_latent_decode_gated_behind_the_map_decode(G, MH, MW, B, seed):
    """lic360_impcodec_decode_masked + lic360_codec_decode_gated: the importance map decodes on one stream and refreshes the latent
    codec's mask after every plane (Dtow(2)(Imp2mask(levels)), lic360_demo.py:283-287), the latent decodes on ANOTHER stream and
    every plane's table kernel waits for the event that covers its positions.  The mask the device derives == the oracle ops'
    mask of the true levels, and the latent comes back exactly -- also when the mask buffer starts out as garbage."""
    import oracle as orc
    from lic360_fused import FusedCodec, FusedImpCodec
    rng = np.random.default_rng(seed)
    nsym, H, W = 49, 2 * MH, 2 * MW
    levels = rng.integers(0, nsym, (B, 1, MH, MW)).astype(np.float32)
    mask = orc.dtow(orc.imp2mask(levels, nsym - 1, 4 * G), 2, True)            # [B, G, H, W]
    assert mask.shape == (B, G, H, W) and 0.05 < mask.mean() < 0.95
    code = np.clip(np.rint(rng.normal(3.5, 1.2, (B, G, H, W))), 0, 7).astype(np.float32)
    ic = FusedImpCodec(MH, MW, max_batch=B, hidden_channels=8, nsym=nsym)
    ic.load_layers(rc.make_imp_params(4000 + seed, 8, nsym))
    fc = FusedCodec(G, H, W, max_batch=B)
    fc.load_layers(rc.make_main_params(2000 + seed, G))
    ic.encode_async(dev(levels))
    fc.encode_async(dev(code), dev(mask))
    torch.cuda.synchronize()
    assert int(ic.err[:B].abs().sum().item()) == 0 and int(fc.err[:B].abs().sum().item()) == 0
    s_map, s_lat = torch.cuda.Stream(), torch.cuda.Stream()
    mbuf = torch.full((B, G, H, W), 7.0, dtype=torch.float32, device="cuda:0")   # garbage: a mask read too early would show
    for _ in range(2):
        ic.levels_out.fill_(48.0)
        torch.cuda.synchronize()
        with torch.cuda.stream(s_map):
            gate = ic.decode_masked_async(B, mbuf, mask_channels=4 * G, stride=2)     # (records its events before the waits below are queued)
        with torch.cuda.stream(s_lat):
            out = fc.decode_async(mbuf, B, gate=gate)
        torch.cuda.synchronize()
        assert int(ic.err[:B].abs().sum().item()) == 0 and int(fc.err[:B].abs().sum().item()) == 0
        assert np.array_equal(ic.levels_out[:B].cpu().numpy(), levels)
        assert np.array_equal(mbuf.cpu().numpy(), mask)
        assert np.array_equal(out.cpu().numpy(), code * mask)
        mbuf.fill_(7.0)
    # the gate is a one-time ticket: a gate used twice, a gate of an earlier step and a foreign mask buffer are errors,
    # not a silent decode against whatever the re-recorded events and the buffer happen to hold
    import lic360
    with pytest.raises(lic360.Lic360Error, match="stale gate"):
        fc.decode_async(mbuf, B, gate=gate)
    with torch.cuda.stream(s_map):
        old_gate = ic.decode_masked_async(B, mbuf, mask_channels=4 * G, stride=2)
        new_gate = ic.decode_masked_async(B, mbuf, mask_channels=4 * G, stride=2)
    with pytest.raises(lic360.Lic360Error, match="stale gate"):
        fc.decode_async(mbuf, B, gate=old_gate)
    with pytest.raises(lic360.Lic360Error, match="not the buffer"):
        fc.decode_async(torch.zeros_like(mbuf), B, gate=new_gate)
    with torch.cuda.stream(s_lat):
        out = fc.decode_async(mbuf, B, gate=new_gate)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), code * mask)


@pytest.mark.parametrize("coder", ["device", "host"])
def test_fused_codec_corrupt_stream_is_reported_not_fatal(coder):
    """A truncated / damaged bitstream must end in an error (or a wrong-but-finite decode), never in a hang or a fault:
    the decoders (device wave or host thread) read zeros past the end of a stream, like the reference's BitInputStream."""
    from lic360_fused import FusedCodec, Lic360Error
    G, H, W, B = 6, 8, 12, 2
    rng = np.random.default_rng(41)
    layers = rc.make_main_params(2041, G)
    items = [latent(rng, G, H, W) for _ in range(B)]
    code = np.concatenate([it[0] for it in items], 0)
    mask = np.concatenate([it[1] for it in items], 0)
    fc = FusedCodec(G, H, W, max_batch=B)
    fc.load_layers(layers)
    fc.set_coder(coder)
    good = fc.encode(dev(code), dev(mask))
    damaged = [good[0][:len(good[0]) // 2], bytes(b ^ 0x5A for b in good[1])]
    try:
        out = fc.decode(damaged, dev(mask)).cpu().numpy()
        assert np.isfinite(out).all() and out.min() >= 0 and out.max() <= 7
    except Lic360Error:
        pass
    # the codec is still usable afterwards
    assert np.array_equal(fc.decode(good, dev(mask)).cpu().numpy(), code * mask)
