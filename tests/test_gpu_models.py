"""The analysis / synthesis transforms on this backend (lic360_models.py): the one-pass GDN against its torch statement,
blocks against index-only torch formulations, the reference's state_dict layout, and the whole codec end to end -- image -> transforms ->
device-resident entropy codecs -> bitstreams -> transforms -> image -- with seeded random weights (no checkpoint exists offline)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lic():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lic360
    return lic360


def _pad_torch(x, pad):
    W = x.shape[-1]
    body = torch.cat([x[..., W - pad:], x, x[..., :pad]], -1)
    def across(r):
        r = torch.flip(r, (-1,))
        return torch.cat([r[..., W - pad:], r, r[..., :pad]], -1)
    return torch.cat([across(torch.flip(x[..., :pad, :], (-2,))), body, across(torch.flip(x[..., x.shape[-2] - pad:, :], (-2,)))], -2)


def _refresh(x, pad=2):
    """SpherePad in place: the apron of a padded map recomputed from its interior"""
    return _pad_torch(x[..., pad:-pad, pad:-pad], pad)


def _trim(x, pad):
    y = torch.zeros_like(x)
    y[..., pad:-pad, pad:-pad] = x[..., pad:-pad, pad:-pad]
    return y


@pytest.mark.parametrize("c,inverse", [(192, False), (192, True), (96, False), (16, True), (48, False)])
def test_fused_gdn_matches_torch(lic, c, inverse):
    g = torch.Generator(device="cuda:0").manual_seed(c)
    x = torch.randn((2, c, 20, 37), device="cuda:0", generator=g)           # 740 positions: a ragged last tile
    gamma = (torch.rand((c, c), device="cuda:0", generator=g) * 0.02 + 0.1 * torch.eye(c, device="cuda:0"))
    beta = torch.rand((c,), device="cuda:0", generator=g) + 0.5
    want = torch.sqrt(F.conv2d(x * x, gamma.view(c, c, 1, 1), beta))
    want = x * want if inverse else x / want
    got = lic.gdn_forward(x, gamma, beta, inverse)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), float((got - want).abs().max())
    want64 = torch.sqrt(torch.einsum("ij,njhw->nihw", gamma.double(), x.double() ** 2) + beta.double().view(1, c, 1, 1))
    want64 = x.double() * want64 if inverse else x.double() / want64
    assert torch.allclose(got.double(), want64, rtol=2e-6, atol=1e-6)                 # against float64, no library in the loop
    import lic360_operator as lo
    m = lo.GDN(c, 0, inverse)
    with torch.no_grad():
        y_fused = m(x)
    y_torch = m(x.clone().requires_grad_(True))                             # the recording path stays on torch
    assert torch.allclose(y_fused, y_torch.detach(), rtol=1e-4, atol=1e-5)


def test_blocks_match_index_only_torch(lic):
    import lic360_models as lm
    torch.manual_seed(3)
    c = 16
    x = _refresh(torch.randn((1, c, 12, 20), device="cuda:0"))              # a map with a valid 2-cell apron
    with torch.no_grad():
        blk = lm.ResidualBlockV2(c, 0).to("cuda:0")
        y = _trim(blk.relu1(blk.conv1(_refresh(x))), 1)
        want = x + _trim(blk.relu2(blk.conv2(y)), 2)
        assert torch.allclose(blk(x.clone()), want, rtol=1e-4, atol=1e-4)
        bt = lm.ResidualBlock(c, 0).to("cuda:0")
        t = _refresh(x)
        want = _trim(x + bt.conv3(bt.relu2(bt.conv2(bt.relu1(bt.conv1(t))))), 2)
        assert torch.allclose(bt(x.clone()), want, rtol=1e-4, atol=1e-4)
        dn = lm.ResidualBlockDown(c, c, 0).to("cuda:0")
        t = _refresh(x)
        yy = _refresh(_trim(dn.relu1(dn.conv1(t)), 2))
        want = _trim(dn.short_cut(x) + dn.relu2(dn.conv2(yy)), 2)
        assert torch.allclose(dn(x.clone()), want, rtol=1e-4, atol=1e-4)
        up = lm.ResidualBlockUp(c, 0).to("cuda:0")
        b = _trim(F.pixel_shuffle(up.relu1(up.conv1(_refresh(x))), 2), 2)
        b = up.relu2(up.conv2(_refresh(b)))
        want = _trim(b + F.pixel_shuffle(up.short_cut(x[..., 1:-1, 1:-1]), 2), 2)
        assert torch.allclose(up(x.clone()), want, rtol=1e-4, atol=1e-4)


def test_block_gradients_match_index_only_torch(lic):
    """the in-place SpherePad / SphereTrim wrappers return their input without mark_dirty (as the reference's do), which is
    only sound if every tensor a convolution saved for backward still holds what it read.  Parameter AND input gradients of the blocks
    that refresh / trim in place (ResidualBlockV2, ResidualBlockDown, the three-deep trunk of AttentionBlock) against the same blocks
    written with index-only, out-of-place torch operations on the same weights."""
    import lic360_models as lm
    torch.manual_seed(9)
    c = 16
    x0 = _refresh(torch.randn((1, c, 12, 20), device="cuda:0"))

    def grads(fn, params):
        x = x0.clone().requires_grad_(True)
        out = fn(x)
        # (a gradient as it arrives in the network: the consumer of a block's output refreshes or trims its apron first, so apron
        # cells carry none -- ResidualBlockV2's skip path hands x's apron through untrimmed, exactly as the reference's does)
        g = _trim(torch.sin(torch.arange(out.numel(), device="cuda:0", dtype=torch.float32)).view_as(out), 2)
        return torch.autograd.grad((out * g).sum(), [x] + params, allow_unused=False)

    v2 = lm.ResidualBlockV2(c, 0).to("cuda:0")
    ref_v2 = lambda x: _refresh(x) + _trim(v2.relu2(v2.conv2(_trim(v2.relu1(v2.conv1(_refresh(x))), 1))), 2)
    dn = lm.ResidualBlockDown(c, c, 0).to("cuda:0")
    gdn = dn.relu2

    def ref_dn(x):
        y = _refresh(_trim(dn.relu1(dn.conv1(_refresh(x))), 2))
        return _trim(dn.short_cut(x) + gdn(dn.conv2(y)), 2)
    rb = lm.ResidualBlock(c, 0).to("cuda:0")
    ref_rb = lambda x: _trim(_refresh(x) + rb.conv3(rb.relu2(rb.conv2(rb.relu1(rb.conv1(_refresh(x)))))), 2)
    for blk, ref in ((v2, ref_v2), (dn, ref_dn), (rb, ref_rb)):
        params = [q for q in blk.parameters()]
        got = grads(lambda x: blk(x * 1.0), params)                          # (x * 1.0: the block works in place on its input)
        want = grads(ref, params)
        for a, b, name in zip(got, want, ["input"] + [n for n, _ in blk.named_parameters()]):
            if name == "input":                                             # interior only: what the reference's in-place pad backward leaves in the
                a, b = a[..., 2:-2, 2:-2], b[..., 2:-2, 2:-2]               # apron cells of its input gradient is its own (an upstream trim zeroes it)
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-4), "%s %s: max abs diff %g" % (type(blk).__name__, name, float((a - b).abs().max()))


def _block_params(blk):
    """numpy parameters of a block under its state_dict keys + the constants of its GDN (the oracle's `blocks` take these)"""
    p = {k: v.detach().cpu().numpy() for k, v in blk.state_dict().items()}
    for name, m in blk.named_modules():
        if type(m).__name__ == "GDN":
            p[name + ".pedestal"], p[name + ".beta_bound"], p[name + ".gamma_bound"] = m.pedestal, m.beta_bound, m.gamma_bound
    return p


def test_blocks_match_the_oracle(lic):
    """ResidualBlock / ResidualBlockV2 / ResidualBlockDown / ResidualBlockUp (test/model_zoo.py:8-95,145-170) against the CPU oracle's
    restatement of the same blocks (oracle.blocks: orc_conv2d / orc_prelu / orc_gdn + the sphere ops' and pixel shuffle's restatements).
    The convolutions are MIOpen's here and cuDNN's in the reference: neither fixes a summation order, so the comparison is 1e-4, not
    bit-exact; everything around them (aprons, trims, shuffles, the residual wiring) is exact and would show as O(1) errors."""
    import oracle as orc
    import lic360_models as lm
    torch.manual_seed(5)
    c = 16
    x = _refresh(torch.randn((2, c, 12, 20), device="cuda:0"))              # maps with a valid 2-cell apron
    xn = x.cpu().numpy()
    with torch.no_grad():
        for cls, fn in ((lm.ResidualBlock, orc.blocks.residual), (lm.ResidualBlockV2, orc.blocks.residual_v2),
                        (lambda ch, d: lm.ResidualBlockDown(ch, ch, d), orc.blocks.residual_down), (lm.ResidualBlockUp, orc.blocks.residual_up)):
            blk = cls(c, 0).to("cuda:0")
            for prm in blk.parameters():                                    # PReLU slopes / GDN parameters away from their symmetric defaults
                if prm.dim() <= 2:
                    prm.add_(0.05 * torch.rand_like(prm))
            got = blk(x.clone()).cpu().numpy()
            want = fn(xn.copy(), _block_params(blk))
            assert got.shape == want.shape, (got.shape, want.shape)
            err = np.abs(got - want).max()
            assert np.allclose(got, want, rtol=1e-4, atol=1e-4), "%s: max abs error %g" % (type(blk).__name__, err)


@pytest.mark.parametrize("case", [(32, 96, 20, 36, 2, 1, 0, True, True, 2), (16, 192, 21, 37, 1, 1, 0, True, False, 1), (48, 192, 12, 20, 1, 0, 1, True, False, 1),
                                  (32, 384, 18, 34, 2, 0, 0, False, True, 2), (64, 96, 9, 70, 1, 1, 0, False, False, 1), (32, 192, 22, 40, 1, 1, 0, True, False, 2),
                                  (16, 192, 38, 24, 2, 2, 0, True, True, 2)],
                         ids=lambda c: "%dto%d_%dx%d_ring%d_%d_sphere%d" % (c[0], c[1], c[2], c[3], c[4], c[9], c[5]))
def test_sconv3x3_matches_the_oracle_conv(lic, case):
    """lic360_sconv3x3 (csrc/conv3x3_kernels.hip: apron by index, bias + PReLU + residual in the epilogue, window = the trim) against the
    oracle's restatement of what it replaces: in-place sphere pad -> conv2d -> PReLU -> (+ residual) on the window; cells outside the window
    are not touched.  Odd sizes (ragged tiles), two output-channel blocks, the unpadded form (crop = 1), plain aprons (sphere = 0), the window
    without the 1-ring columns (ring_w = 2) and the longitude-wrap-only read (sphere = 2) that ResidualBlockV2's second convolution uses, a
    window of 16 k + 2 rows (the tall last tile row)."""
    import oracle as orc
    cin, cout, hp, wp, ring, sphere, crop, act, with_res, ring_w = case
    rng = np.random.default_rng(cin + 7 * cout + hp)
    x = rng.standard_normal((2, cin, hp, wp)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    b, sl = rng.standard_normal(cout).astype(np.float32), rng.random(cout).astype(np.float32)
    res = rng.standard_normal((2, cout, hp, wp)).astype(np.float32) if with_res else None
    xin = x
    if sphere == 1:
        xin = orc.sphere_pad_inplace(x.copy(), 2)
    elif sphere == 2:                                                         # columns of the apron from the interior, rows as they are
        xin = x.copy()
        xin[..., :2], xin[..., wp - 2:] = x[..., wp - 4:wp - 2], x[..., 2:4]
    want = orc.conv2d(xin, w, b, 1, 1 - crop)
    if act:
        want = orc.prelu(want, sl)
    dev = lambda t: None if t is None else torch.from_numpy(t).cuda()
    out = torch.full((2, cout, hp - 2 * crop, wp - 2 * crop), 7.0, device="cuda:0")
    lic.sconv3x3(dev(x), lic.sconv3x3_pack(dev(w)), dev(b), dev(sl) if act else None, dev(res), out, pad=2, sphere=sphere, ring=ring, crop=crop, ring_w=ring_w)
    got = out.cpu().numpy()
    win = (slice(None), slice(None), slice(ring - crop, hp - crop - ring), slice(ring_w - crop, wp - crop - ring_w))
    if with_res:
        want = want + res
    assert np.allclose(got[win], want[win], rtol=1e-4, atol=1e-4), float(np.abs(got[win] - want[win]).max())
    frame = np.ones(got.shape, bool)
    frame[win] = False
    assert np.all(got[frame] == 7.0)


def test_sconv3x3_with_the_pixel_shuffle_as_its_store_pattern(lic):
    """the unpadded 3x3 conv -> PReLU -> Dtow(2) chain of ResidualBlockUp (test/model_zoo.py:160-162) in one launch: the shuffled window equals the
    oracle's sphere pad -> conv2d -> PReLU -> dtow; cells of the shuffled map outside the window are not touched"""
    import oracle as orc
    cin, cout, hp, wp = 32, 192, 14, 22
    rng = np.random.default_rng(77)
    x = rng.standard_normal((2, cin, hp, wp)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    b, sl = rng.standard_normal(cout).astype(np.float32), rng.random(cout).astype(np.float32)
    want = orc.dtow(orc.prelu(orc.conv2d(orc.sphere_pad_inplace(x.copy(), 2), w, b, 1, 0), sl), 2, True)       # [2, 48, 2 (hp - 2), 2 (wp - 2)]
    dev = lambda t: torch.from_numpy(t).cuda()
    out = torch.full((2, cout // 4, 2 * (hp - 2), 2 * (wp - 2)), 7.0, device="cuda:0")
    lic.sconv3x3(dev(x), lic.sconv3x3_pack(dev(w)), dev(b), dev(sl), None, out, pad=2, sphere=1, ring=2, crop=1, shuffle=True)
    got = out.cpu().numpy()
    assert got.shape == want.shape
    win = (slice(None), slice(None), slice(2, 2 * (hp - 2) - 2), slice(2, 2 * (wp - 2) - 2))      # the shuffled map's interior
    assert np.allclose(got[win], want[win], rtol=1e-4, atol=1e-4), float(np.abs(got[win] - want[win]).max())
    frame = np.ones(got.shape, bool)
    frame[win] = False
    assert np.all(got[frame] == 7.0)


@pytest.mark.parametrize("case", [(64, 96, 20, 36, 2, 2, True, False), (96, 192, 21, 37, 2, 2, False, True), (32, 384, 12, 20, 1, 3, True, True)],
                         ids=lambda c: "%dto%d_%dx%d" % (c[0], c[1], c[2], c[3]))
def test_sconv1x1_matches_the_oracle_conv(lic, case):
    """lic360_sconv1x1 (the transforms' 1x1 layers on the 3x3 kernel's body) against the oracle's conv2d -> PReLU -> + residual on the window;
    cells outside the window are not touched"""
    import oracle as orc
    cin, cout, hp, wp, ring, ring_w, act, with_res = case
    rng = np.random.default_rng(3 * cin + cout + wp)
    x = rng.standard_normal((2, cin, hp, wp)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 1, 1)) * 0.1).astype(np.float32)
    b, sl = rng.standard_normal(cout).astype(np.float32), rng.random(cout).astype(np.float32)
    res = rng.standard_normal((2, cout, hp, wp)).astype(np.float32) if with_res else None
    want = orc.conv2d(x, w, b, 1, 0)
    if act:
        want = orc.prelu(want, sl)
    if with_res:
        want = want + res
    dev = lambda t: None if t is None else torch.from_numpy(t).cuda()
    out = torch.full((2, cout, hp, wp), 7.0, device="cuda:0")
    lic.sconv1x1(dev(x), lic.sconv1x1_pack(dev(w)), dev(b), dev(sl) if act else None, dev(res), out, ring=ring, ring_w=ring_w)
    got = out.cpu().numpy()
    win = (slice(None), slice(None), slice(ring, hp - ring), slice(ring_w, wp - ring_w))
    assert np.allclose(got[win], want[win], rtol=1e-4, atol=1e-4), float(np.abs(got[win] - want[win]).max())
    frame = np.ones(got.shape, bool)
    frame[win] = False
    assert np.all(got[frame] == 7.0)


def test_fused_blocks_match_the_oracle_at_full_width(lic, monkeypatch):
    """the transform blocks at the reference's width (192 channels), where their 3x3 stride-1 convolutions run on lic360.sconv3x3 (forced
    here for a small map), against the oracle's blocks: the whole output, aprons included"""
    import oracle as orc
    import lic360_models as lm
    monkeypatch.setattr(lm, "FUSED_MIN_WORKGROUPS", 0)
    monkeypatch.setattr(lm, "FUSED_MIN_FILL", 0.0)
    torch.manual_seed(6)
    c = 192
    x = _refresh(torch.randn((1, c, 12, 20), device="cuda:0")).contiguous()
    xn = x.cpu().numpy()
    calls = []
    real = lic.sconv3x3
    monkeypatch.setattr(lic, "sconv3x3", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        for cls, fn, ncalls in ((lm.ResidualBlock, orc.blocks.residual, 1), (lm.ResidualBlockV2, orc.blocks.residual_v2, 2),
                                (lambda ch, d: lm.ResidualBlockDown(ch, ch, d), orc.blocks.residual_down, 1), (lm.ResidualBlockUp, orc.blocks.residual_up, 2)):
            blk = cls(c, 0).to("cuda:0")
            for prm in blk.parameters():
                if prm.dim() <= 2:
                    prm.add_(0.05 * torch.rand_like(prm))
            del calls[:]
            got = blk(x.clone()).cpu().numpy()
            assert len(calls) == ncalls, (type(blk).__name__, len(calls))
            want = fn(xn.copy(), _block_params(blk))
            assert np.allclose(got, want, rtol=1e-4, atol=1e-4), "%s: max abs error %g" % (type(blk).__name__, np.abs(got - want).max())
        att = lm.AttentionBlock(c, 0).to("cuda:0")
        monkeypatch.setattr(lm, "FUSED_MIN_WORKGROUPS", 1 << 30)
        want = att(x.clone())                                                   # the library path of the same module
        monkeypatch.setattr(lm, "FUSED_MIN_WORKGROUPS", 0)
        del calls[:]
        got = att(x.clone())
        assert len(calls) == 6 and torch.allclose(got, want, rtol=1e-4, atol=1e-4), float((got - want).abs().max())


def test_state_dict_layout_is_the_references(lic):
    import lic360_models as lm
    enc, dec = lm.CMP_Encoder(32, 32, 8, 0), lm.CMP_Decoder(32, 32, 8, 0)
    ek, dk = set(enc.state_dict()), set(dec.state_dict())
    for k in ("encoder.net.0.conv1.weight", "encoder.net.0.relu2.gamma", "encoder.net.0.short_cut.bias", "encoder.net.1.relu1.weight",
              "encoder.net.3.trunk.2.conv3.weight", "encoder.net.3.attention.3.weight", "encoder.net.7.conv.weight", "encoder.net2.1.weight",
              "encoder.imp_net.2.bias", "encoder.imp_net.5.net.0.weight", "encoder.imp_net.5.data", "quant.weight", "quant.count"):
        assert k in ek, k
    for k in ("decoder.net.0.conv.weight", "decoder.net.1.trunk.0.conv1.weight", "decoder.net.3.conv1.weight", "decoder.net.3.relu2.beta",
              "decoder.net.3.short_cut.weight", "decoder.net.11.weight", "quant.weight"):
        assert k in dk, k
    assert tuple(enc.state_dict()["encoder.net.0.conv1.weight"].shape) == (32, 3, 3, 3)
    assert tuple(dec.state_dict()["decoder.net.3.conv1.weight"].shape) == (128, 32, 3, 3)


def test_whole_codec_end_to_end(lic):
    """image -> analysis -> fused entropy codecs (latent + importance map) -> bytes -> decode -> synthesis -> image: what comes out of
    the bitstreams is exactly what went in, so the reconstruction equals the one computed from the encoder-side symbols"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import lic360_models as lm
    from lic360_fused import FusedCodec, FusedImpCodec
    from util import make_main_params, make_imp_params
    torch.manual_seed(11)
    C = 32                                                                  # 8 groups of 4: the codec's structure at a fifth of the width
    enc, dec = lm.CMP_Encoder(C, C, 8, 0).to("cuda:0").eval(), lm.CMP_Decoder(C, C, 8, 0).to("cuda:0").eval()
    with torch.no_grad():
        dec.quant.weight.copy_(enc.quant.weight)
        img = torch.rand((2, 3, 512, 1024), device="cuda:0")
        code, mask, levels = enc(img)
    G = C // 4
    assert tuple(code.shape) == (2, G, 64, 128) and tuple(mask.shape) == (2, G, 64, 128) and tuple(levels.shape) == (2, 1, 32, 64)
    assert float(code.min()) >= 0 and float(code.max()) <= 7 and bool(((mask == 0) | (mask == 1)).all())
    assert float(levels.min()) >= 0 and float(levels.max()) <= G and bool((levels == torch.round(levels)).all())
    # the mask is the importance level unrolled over the groups (what Imp2mask rebuilds on the decoder side)
    lv_up = levels.repeat_interleave(2, 2).repeat_interleave(2, 3)
    assert torch.equal(mask, (torch.arange(G, device="cuda:0").view(1, G, 1, 1) < lv_up).float())
    fc = FusedCodec(G, 64, 128, max_batch=2)
    fc.load_layers(make_main_params(5, G))
    ic = FusedImpCodec(32, 64, max_batch=2, hidden_channels=3 * G, nsym=G + 1)
    ic.load_layers(make_imp_params(5, cpg=3 * G, nsym=G + 1))
    streams, istreams = fc.encode(code.contiguous(), mask.contiguous()), ic.encode(levels.contiguous())
    assert all(len(s) > 0 for s in streams) and all(len(s) > 0 for s in istreams)
    lv2 = ic.decode(istreams)
    assert torch.equal(lv2, levels)
    mask2 = (torch.arange(G, device="cuda:0").view(1, G, 1, 1) < lv2.repeat_interleave(2, 2).repeat_interleave(2, 3)).float()
    code2 = fc.decode(streams, mask2)
    assert torch.equal(code2, code * mask)
    with torch.no_grad():
        rec, rec_ref = dec(code2, mask2), dec(code * mask, mask)
    assert tuple(rec.shape) == (2, 3, 512, 1024) and torch.equal(rec, rec_ref) and bool(torch.isfinite(rec).all())
