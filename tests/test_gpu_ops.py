"""GPU parity tests: every op of the drop-in `lic360` module (HIP, through the C ABI) against the
CPU oracle on the same seeded inputs.  Integer / index / copy work and the masked convolution are
required to be BIT-EXACT (np.array_equal); EntropyGmm (log-likelihood) is checked at 1e-5 as
BASELINE.json's north_star states."""
import numpy as np
import pytest
import torch

import oracle as orc
from util import conv_params, latent, case_rng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lic():
    import lic360
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return lic360


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ masked convolution, encode order
CONV_CASES = [
    # ngroup, cin, cout, hidden, act, nb, N, H, W
    (6, 1, 4, False, True, 3, 3, 8, 12),       # first layer of the latent net, small
    (6, 4, 4, True, True, 3, 3, 8, 12),        # hidden layer
    (6, 4, 3, True, False, 3, 3, 7, 19),       # last layer (cout=3), ragged H/W
    (48, 1, 4, False, True, 3, 3, 4, 16),      # full channel count, first
    (48, 4, 4, True, True, 3, 3, 8, 16),       # full channel count, hidden: 126 KB LDS tile
    (48, 4, 3, True, False, 3, 6, 4, 16),      # last layer, two samples per net
    (1, 1, 20, False, True, None, 1, 9, 11),   # importance-map net first layer (no batch)
    (1, 8, 20, True, True, None, 2, 6, 10),    # 200 tap indices -> multi-index lanes
    (1, 20, 5, True, False, None, 1, 5, 7),    # nout not a multiple of 16
    (1, 144, 49, True, False, None, 1, 4, 6),  # importance-map net last layer, real channel counts
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "g%d_%dto%d_%s" % (c[0], c[1], c[2], "h" if c[3] else "f"))
def test_cconv_ec_bit_exact(lic, case):
    G, cin, cout, hidden, act, nb, N, H, W = case
    rng = case_rng(case)
    C, nout = G * cin, G * cout
    w, b, a = conv_params(rng, nb, nout, C, act=act)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    x[rng.random(x.shape) < 0.2] = 0.0
    constrain = 6 if hidden else 5
    ref = orc.cconv_ec(x, w, b, a, G, constrain)
    op = lic.CconvEcOp(C, G, nout, 5, constrain, 0, False)
    args = [dev(x), dev(w), dev(b)] + ([dev(a)] if act else [])
    name = ("forward_act" if act else "forward") + ("_batch" if nb else "")
    got = host(getattr(op, name)(*args)[0])
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), "max abs diff %g" % np.abs(got - ref).max()


def test_cconv_ec_matches_masked_conv2d(lic):
    """Independent pin: equals F.conv2d with the mask_constrain rule (extension/mask_constrain_cuda.cu:17-41)."""
    rng = np.random.default_rng(7)
    G, cin, cout, H, W = 5, 4, 4, 6, 7
    for hidden in (False, True):
        C, nout = G * cin, G * cout
        w, b, _ = conv_params(rng, None, nout, C, act=False)
        x = rng.standard_normal((2, C, H, W)).astype(np.float32)
        o, ti, kh, kw = np.meshgrid(np.arange(nout), np.arange(C), np.arange(5), np.arange(5), indexing="ij")
        keep = (kh + kw + ti // cin < o // cout + 4) if not hidden else (kh + kw + ti // cin <= o // cout + 4)
        ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w * keep).double(),
                                         torch.from_numpy(b).double(), padding=2).numpy()
        op = lic.CconvEcOp(C, G, nout, 5, 6 if hidden else 5, 0, False)
        got = host(op.forward(dev(x), dev(w), dev(b))[0])
        assert np.abs(got - ref).max() < 1e-4


# ------------------------------------------------------------------ masked convolution, decode order
@pytest.mark.parametrize("case", [(6, 1, 4, False, True, 3, 3, 6, 9), (6, 4, 4, True, True, 3, 3, 6, 9),
                                  (6, 4, 3, True, False, 3, 3, 5, 8), (1, 1, 20, False, True, None, 1, 6, 7),
                                  (1, 20, 7, True, False, None, 1, 5, 6)],
                         ids=lambda c: "g%d_%dto%d" % (c[0], c[1], c[2]))
def test_cconv_dc_planes_bit_exact(lic, case):
    G, cin, cout, hidden, act, nb, N, H, W = case
    rng = case_rng(case)
    C, nout = G * cin, G * cout
    w, b, a = conv_params(rng, nb, nout, C, act=act)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    constrain = 6 if hidden else 5
    idx, pidx = orc.code_contex(H, W)
    ctx = lic.CodeContexOp(0, False)
    p1, p2 = ctx.forward(dev(x))
    assert np.array_equal(host(p1).reshape(-1), np.stack([idx[:H * W].reshape(H, W), idx[H * W:].reshape(H, W)], -1).reshape(-1)) \
        or np.array_equal(host(p1).reshape(-1), idx)
    assert np.array_equal(p2.numpy(), pidx)
    op = lic.CconvDcOp(C, G, nout, 5, constrain, 0, False)
    op.set_param(p1, p2)
    op.restart()
    ref = np.zeros((N, nout, H, W), np.float32)
    args = [dev(x), dev(w), dev(b)] + ([dev(a)] if act else [])
    name = ("forward_act" if act else "forward") + ("_batch" if nb else "")
    for p in range(H + W + G - 2):
        orc.cconv_dc_plane(x, w, b, a, ref, G, constrain, idx, pidx, p)
        got = host(getattr(op, name)(*args)[0])
        assert np.array_equal(got, ref), "plane %d" % p
    # SURVEY.md §4 invariant: the final DC buffer equals the EC output (up to the sign of zero)
    ec = orc.cconv_ec(x, w, b, a, G, constrain)
    assert np.array_equal(ref, ec)


# ------------------------------------------------------------------ plane gather / scatter / add
def test_tile_ops(lic):
    rng = np.random.default_rng(3)
    G, cpn, H, W = 5, 3, 6, 8
    y = rng.standard_normal((3, G * cpn, H, W)).astype(np.float32)
    code = rng.integers(0, 8, (1, G, H, W)).astype(np.float32)
    idx, pidx = orc.code_contex(H, W)
    p1, p2 = lic.CodeContexOp(0, False).forward(dev(code))
    ext_b, ext, ext_nl = lic.TileExtractOp(G, True, 0, False), lic.TileExtractOp(G, True, 0, False), lic.TileExtractOp(G, False, 0, False)
    ipt, add = lic.TileInputOp(G, -3.5, 1.0, 3, 0, False), lic.TileAddOp(G, 0, False)
    for o in (ext_b, ext, ext_nl, ipt, add):
        o.set_param(p1, p2)
        o.restart()
    ya, yb = rng.standard_normal((3, G * 4, H, W)).astype(np.float32), rng.standard_normal((3, G * 4, H, W)).astype(np.float32)
    ya_d, yb_d = dev(ya), dev(yb)
    ob = np.zeros((3, cpn, H, W), np.float32)
    ol = np.zeros((1, 1, H, W), np.float32)
    onl = np.zeros((1, 1, H, W), np.float32)
    oin = np.zeros((3, G, H, W), np.float32)
    sym = np.zeros((1, 1, H, W), np.float32)
    for p in range(H + W + G):            # two steps past the last plane on purpose
        nb_ = orc.tile_extract_batch(y, ob.reshape(-1), G, idx, pidx, p)
        z, le = ext_b.forward_batch(dev(y))
        assert int(le[0]) == nb_
        zz, st = host(z).reshape(-1), cpn * H * W                     # only the slab prefixes are defined (op-owned at::empty buffer)
        for sl in range(3):
            assert np.array_equal(zz[sl * st: sl * st + nb_ * cpn], ob.reshape(-1)[sl * st: sl * st + nb_ * cpn])
        nl = orc.tile_extract(code, ol.reshape(-1), G, True, idx, pidx, p)
        z, le = ext.forward(dev(code))
        assert int(le[0]) == nl and np.array_equal(host(z).reshape(-1)[:nl], ol.reshape(-1)[:nl])
        orc.tile_extract(code, onl.reshape(-1), G, False, idx, pidx, p)
        z, _ = ext_nl.forward(dev(code))
        assert np.array_equal(host(z), onl)
        sym.reshape(-1)[:] = rng.integers(0, 8, H * W)
        orc.tile_input(sym.reshape(-1), oin.reshape(-1), 1, G, H, W, -3.5, 1.0, 3, idx, pidx, p)
        assert np.array_equal(host(ipt.forward(dev(sym))[0]), oin)
        if p < H + W + G - 2:
            orc.tile_add(ya, yb, G, idx, pidx, p)
            add.forward(ya_d, yb_d)
            assert np.array_equal(host(ya_d), ya)


# ------------------------------------------------------------------ CDF tables
def test_gmm_table_batch_bit_exact(lic):
    rng = np.random.default_rng(11)
    H, W = 16, 32
    tn = 300
    data = np.zeros((3, 3, H, W), np.float32)
    flat = data.reshape(-1)
    stride = flat.size // 3
    flat[:tn * 3] = rng.standard_normal(tn * 3) * 2                      # mixture logits
    flat[stride:stride + tn * 3] = rng.uniform(-0.3, 2.5, tn * 3)        # sigma (some negative -> beta floor)
    flat[2 * stride:2 * stride + tn * 3] = rng.uniform(-5, 5, tn * 3)    # mu (re-centred symbols live in [-3.5,3.5])
    flat[stride:stride + 6] = [1e-6, -1.0, 0.0, 3e-4, 50.0, 1e-3]       # degenerate sigmas -> fix-up path
    ref_in = data.copy()
    ref = orc.gmm_table_batch(ref_in.reshape(-1), stride, tn)
    op = lic.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6, 0, False)
    d = dev(data)
    got = host(op.forward_batch(d, torch.tensor([tn], dtype=torch.int32))[0])
    assert got.shape == (3 * H * W * 3 // 3 // 3 * 3 // 3, 9) or got.shape[1] == 9
    assert np.array_equal(got[:tn], ref)
    assert np.array_equal(host(d), ref_in)          # in-place softmax / sigma floor, like the reference
    assert np.all(np.diff(got[:tn], axis=1) > 0) and np.all(got[:tn, 0] == 0) and np.all(got[:tn, 8] == 65536)
    # non-batch entry point
    w, dl, m = (np.ascontiguousarray(ref_in.reshape(3, -1)[i][:tn * 3].reshape(1, 1, tn, 3)) for i in range(3))
    w0, d0, m0 = (data.reshape(3, -1)[i][:tn * 3].reshape(1, 1, tn, 3).copy() for i in range(3))
    got2 = host(op.forward(dev(w0), dev(d0), dev(m0), torch.tensor([tn], dtype=torch.int32))[0])
    assert np.array_equal(got2[:tn], ref)


def test_gmm_table_close_to_float64(lic):
    from scipy import special
    rng = np.random.default_rng(12)
    tn = 500
    w = rng.standard_normal((tn, 3))
    s = rng.uniform(0.05, 3, (tn, 3))
    m = rng.uniform(-4, 4, (tn, 3))
    sw = np.exp(w - w.max(1, keepdims=True))
    sw /= sw.sum(1, keepdims=True)
    edges = np.arange(1, 8) - 4.0
    cdf = (sw[:, None, :] * (0.5 + 0.5 * special.erf((edges[None, :, None] - m[:, None, :]) / (s[:, None, :] + 1e-6) / np.sqrt(2)))).sum(-1)
    exact = np.floor(65536 * cdf + 0.5)
    op = lic.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6, 0, False)
    mk = lambda a: dev(a.astype(np.float32).reshape(1, 1, tn, 3))
    got = host(op.forward(mk(w), mk(s), mk(m), torch.tensor([tn], dtype=torch.int32))[0])[:tn]
    well = np.all(np.diff(exact, axis=1) > 2, axis=1) & (exact[:, 0] > 2) & (exact[:, -1] < 65533)   # rows the fix-up leaves alone
    assert np.abs(got[well, 1:8] - exact[well]).max() <= 1


def test_entropy_table_bit_exact(lic):
    rng = np.random.default_rng(13)
    n = 37
    logits = (rng.standard_normal((1, 49, 8, 8)) * 3).astype(np.float32)
    logits.reshape(-1)[:49] = -40.0
    logits.reshape(-1)[5] = 10.0            # one-hot row -> many zero-width bins -> fix-up
    ref = orc.entropy_table(logits.reshape(-1)[:n * 49].copy(), n, 49)
    op = lic.EntropyTableOp(49, 65536, 0, False)
    got = host(op.forward(dev(logits), torch.tensor([n], dtype=torch.int32))[0])
    assert got.shape == (64, 50)
    assert np.array_equal(got[:n], ref)
    assert np.all(np.diff(ref, axis=1) > 0) and np.all(ref[:, -1] == 65536)


def test_entropy_gmm_loss(lic):
    from scipy import special
    rng = np.random.default_rng(14)
    M = 4000
    w = rng.random((M, 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    d = rng.uniform(0.1, 3, (M, 3)).astype(np.float32)
    m = rng.uniform(-3, 3, (M, 3)).astype(np.float32)
    lab = rng.integers(-3, 4, (M, 1)).astype(np.float32)
    op = lic.EntropyGmmOp(3, -1, 0, False)
    loss = host(op.forward(dev(w), dev(d), dev(m), dev(lab))[0])
    ref = orc.entropy_gmm(w, d, m, lab)
    assert np.array_equal(loss, ref[0])                                   # same routines -> identical
    phi = lambda z: 0.5 + 0.5 * special.erf(z / np.sqrt(2))
    p = (w.astype(np.float64) * (phi((lab + 0.5 - m) / d) - phi((lab - 0.5 - m) / d))).sum(1)
    big = p > 0.02            # fp32 evaluation of Phi_b - Phi_a carries ~6e-8 absolute error -> 1e-5 on the log needs p >~ 0.01
    assert big.sum() > M // 2 and np.abs(loss - (-np.log(p + 1e-7)))[big].max() < 1e-5
    grads = op.backward(dev(np.ones(M, np.float32)))
    for g, r in zip(grads, ref[1:]):
        assert np.allclose(host(g), r, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ sphere / pointwise / layout
@pytest.mark.parametrize("shape,pad", [((1, 3, 16, 32), 2), ((2, 5, 9, 14), 2), ((1, 2, 6, 6), 1), ((1, 4, 8, 8), 3)])
def test_sphere_ops(lic, shape, pad):
    rng = np.random.default_rng(21)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = orc.sphere_pad(x, pad)
    assert np.array_equal(host(lic.SpherePadOp(pad, False, 0, False).forward(dev(x))[0]), ref)
    # independent pin: roll/flip formulation of the ERP border
    H, W = shape[2:]
    mid = np.concatenate([x[..., -pad:], x, x[..., :pad]], -1)
    top = mid[:, :, :pad][:, :, ::-1, ::-1]
    bot = mid[:, :, -pad:][:, :, ::-1, ::-1]
    assert np.array_equal(ref, np.concatenate([top, mid, bot], 2))
    # in place: corrupt the apron, refresh it
    y = ref.copy()
    y[:, :, :pad] = 7; y[:, :, -pad:] = 7; y[..., :pad] = 7; y[..., -pad:] = 7
    yd = dev(y)
    out = lic.SpherePadOp(pad, True, 0, False).forward(yd)[0]
    assert out.data_ptr() == yd.data_ptr() and np.array_equal(host(yd), ref)
    assert np.array_equal(orc.sphere_pad_inplace(y.copy(), pad), ref)
    td = dev(ref)
    lic.SphereTrimOp(pad, 0, False).forward(td)
    assert np.array_equal(host(td), orc.sphere_trim(ref.copy(), pad))
    assert np.array_equal(host(lic.SphereCutEdgeOp(pad, 0, False).forward(dev(ref))[0]), x)
    if shape[2] % 4 == 0:
        wgt = rng.random((1, 1, shape[2] // 4)).astype(np.float32)
        got = host(lic.SphereLatScaleOp(shape[2] // 4, 0, False).forward(dev(x), dev(wgt))[0])
        assert np.array_equal(got, orc.sphere_lat_scale(x, wgt, shape[2] // 4))


def test_dtow_impmap_quant(lic):
    rng = np.random.default_rng(22)
    N, C, H, W, levels = 2, 24, 6, 10, 6
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    up = host(lic.DtowOp(2, True, 0, False).forward(dev(x))[0])
    assert np.array_equal(up, orc.dtow(x, 2, True))
    assert np.array_equal(up, torch.nn.functional.pixel_shuffle(torch.from_numpy(x), 2).numpy())
    assert np.array_equal(host(lic.DtowOp(2, False, 0, False).forward(dev(up))[0]), x)
    imp = np.floor(rng.random((N, 1, H, W)) * levels).astype(np.float32) / levels
    op = lic.ImpMapOp(levels, 0.1, 1.0, 0.5, 0.61, 0.61, 0, 3, 0, False)
    out = op.forward(dev(x), dev(imp))
    ro, rm = orc.imp_map(x, imp, levels)
    assert np.array_equal(host(out[0]), ro) and np.array_equal(host(out[2]), rm)
    assert np.allclose(host(out[1]), orc.imp_map_constrain(N, H, 0.5, 0.61), atol=1e-6)
    lv = np.floor(rng.random((N, 1, H, W)) * (levels + 1)).astype(np.float32)
    assert np.array_equal(host(lic.Imp2maskOp(levels, C, 0, False).forward(dev(lv))[0]), orc.imp2mask(lv, levels, C))
    assert np.array_equal(host(lic.ScaleOp(-1.0, 2.0 / 47, 0, False).forward(dev(lv))[0]), orc.scale(lv, -1.0, np.float32(2.0 / 47)))
    wb = np.concatenate([rng.uniform(-1, 0, (C, 1)), rng.uniform(-2, -0.5, (C, 7))], 1).astype(np.float32)
    q = lic.QuantOp(C, 8, 0.9, 100, 2, 0.1, 0, False)
    top, qidx = q.forward(dev(x), dev(wb), dev(np.zeros((C, 8), np.float32)), False)
    rt, rq, rc = orc.quant(x, wb)
    assert np.array_equal(host(top), rt) and np.array_equal(host(qidx), rq) and np.array_equal(host(q.count_data_), rc)
    msk = (rng.random(x.shape) > 0.3).astype(np.float32)
    assert np.array_equal(host(lic.DquantOp(C, 8, 0, False).forward(dev(rq), dev(msk), dev(wb))[0]), orc.dquant(rq, msk, wb))


@pytest.mark.parametrize("levels,N,C,H,W", [(8, 3, 5, 8, 12), (2, 2, 3, 4, 4), (3, 1, 7, 6, 10), (5, 2, 4, 5, 7), (8, 1, 2, 40, 64), (12, 2, 3, 4, 8)])
def test_quant_levels_and_ranges(lic, levels, N, C, H, W):
    """the nearest-centre search for every alphabet size (<= 8 levels: increments in registers, no data-dependent loop; more: the
    generic kernels), 16-byte aligned and unaligned planes, values below the first centre, between centres and beyond the last one"""
    rng = np.random.default_rng(100 * levels + H)
    wb = np.concatenate([rng.uniform(-1, 0, (C, 1)), rng.uniform(-2, 0.3, (C, levels - 1))], 1).astype(np.float32)
    span = float(np.exp(wb[:, 1:]).sum(1).max()) + 2.0
    x = rng.uniform(-2.0, span, (N, C, H, W)).astype(np.float32)
    x.reshape(-1)[::7] = np.repeat(wb[:, 0], N * H * W).reshape(C, -1).T.reshape(-1)[:x.size][::7]      # exact centres too
    q = lic.QuantOp(C, levels, 0.9, 100, 2, 0.1, 0, False)
    top, qidx = q.forward(dev(x), dev(wb), dev(np.zeros((C, levels), np.float32)), False)
    rt, rq, rc = orc.quant(x, wb)
    assert np.array_equal(host(top), rt) and np.array_equal(host(qidx), rq) and np.array_equal(host(q.count_data_), rc)


@pytest.mark.parametrize("ngroup,c_in,c_out,k,hidden", [(6, 4, 4, 5, True), (6, 1, 4, 5, False), (1, 8, 5, 5, True), (4, 3, 2, 3, False), (48, 4, 3, 5, True)])
def test_mask_constrain_and_maskconv2(lic, ngroup, c_in, c_out, k, hidden):
    """MaskConstrainOp zeroes exactly the taps the oracle's rule zeroes (forward on a weight, backward on a gradient, in place), and
    MaskConv2 -- torch's own conv2d over the masked weight, the training-time form of the context model -- agrees with the CconvEc
    kernel on the same weights (different summation order: 1e-4)."""
    import lic360_operator as lo
    rng = np.random.default_rng(ngroup * 10 + k)
    constrain = 6 if hidden else 5
    w = rng.standard_normal((c_out * ngroup, c_in * ngroup, k, k)).astype(np.float32)
    op = lic.MaskConstrainOp(constrain, ngroup, 0, False)
    wd = dev(w)
    assert op.forward(wd) is None and np.array_equal(host(wd), orc.mask_constrain(w, ngroup, constrain))
    gd = dev(w[::-1].copy())
    op.backward(gd)
    assert np.array_equal(host(gd), orc.mask_constrain(w[::-1].copy(), ngroup, constrain))
    if k != 5:
        return                                                              # the context kernels are 5 x 5
    mc = lo.MaskConv2(ngroup, c_in, c_out, k, hidden, 0).to("cuda:0")
    x = dev(rng.standard_normal((2, c_in * ngroup, 9, 13)).astype(np.float32))
    with torch.no_grad():
        mc.bias.copy_(dev(rng.standard_normal(c_out * ngroup).astype(np.float32)))
        y = mc(x)
        assert np.array_equal(host(mc.weight), orc.mask_constrain(host(mc.weight), ngroup, constrain))      # masked in place, idempotent
        ec = lo.CconvEc(ngroup, c_in, c_out, k, hidden, False, 0).to("cuda:0")
        ec.weight.copy_(mc.weight)
        ec.bias.copy_(mc.bias)
        z = ec(x)
    assert torch.allclose(y, z, rtol=1e-4, atol=1e-4), float((y - z).abs().max())
    # (as in the reference, the mask is applied to `weight.data`: dead taps do receive gradients and are zeroed again by the next forward)
    mc.zero_grad()
    mc(x).sum().backward()
    with torch.no_grad():
        mc.weight.add_(mc.weight.grad, alpha=-0.1)
        mc(x)
    assert np.array_equal(host(mc.weight), orc.mask_constrain(host(mc.weight), ngroup, constrain))


def _sphere_pad_torch(x, pad):
    """SpherePad written with torch indexing only (differentiable): wrap columns, mirror + half-turn shifted pole rows"""
    W = x.shape[-1]
    body = torch.cat([x[..., W - pad:], x, x[..., :pad]], -1)                # wrap in longitude
    top = torch.flip(x[..., :pad, :], (-2,))                                # rows pad-1 .. 0, seen from across the pole:
    bot = torch.flip(x[..., x.shape[-2] - pad:, :], (-2,))
    def across(r):                                                          # column tw of the apron shows column W-1-tw, corners likewise
        r = torch.flip(r, (-1,))
        return torch.cat([r[..., W - pad:], r, r[..., :pad]], -1)
    return torch.cat([across(top), body, across(bot)], -2)


@pytest.mark.parametrize("N,C,H,W,pad", [(2, 3, 6, 10, 2), (1, 2, 4, 8, 1), (1, 1, 5, 7, 2)])
def test_sphere_op_gradients(lic, N, C, H, W, pad):
    """SpherePadOp.backward (both forms) and SphereCutEdgeOp.backward: bit-exact against the oracle, and equal to torch autograd
    through an index-only formulation of the forward (which the forward itself is checked against first)"""
    rng = np.random.default_rng(H * 10 + W)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    g = rng.standard_normal((N, C, H + 2 * pad, W + 2 * pad)).astype(np.float32)
    op = lic.SpherePadOp(pad, False, 0, False)
    assert np.array_equal(host(op.forward(dev(x))[0]), host(_sphere_pad_torch(dev(x), pad)))
    got = host(op.backward(dev(g))[0])
    assert np.array_equal(got, orc.sphere_pad_backward(g, pad))
    xt = dev(x).requires_grad_(True)
    _sphere_pad_torch(xt, pad).backward(dev(g))
    assert torch.allclose(dev(got), xt.grad, rtol=1e-6, atol=1e-6)         # same terms, autograd's own summation order
    gi = dev(g)
    assert lic.SpherePadOp(pad, True, 0, False).backward(gi)[0] is gi and np.array_equal(host(gi), orc.sphere_pad_backward_inplace(g.copy(), pad))
    assert np.array_equal(host(gi)[..., pad:-pad, pad:-pad], got)          # interior = the not-in-place gradient, apron untouched
    ce = lic.SphereCutEdgeOp(pad, 0, False)
    gc = rng.standard_normal((N, C, H, W)).astype(np.float32)
    want = np.pad(gc, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    ce.forward(dev(g))
    assert np.array_equal(host(ce.backward(dev(gc))[0]), want) and np.array_equal(orc.sphere_cut_edge_backward(gc, pad), want)


@pytest.mark.parametrize("imp_kernel", [0, 1, 2, 3])
def test_imp_map_backward(lic, imp_kernel):
    """ImpMapOp.backward: gradient of the data under the importance mask and the four importance-gradient rules (v1..v4), bit-exact vs
    the oracle given the same alpha_t; alpha_t itself (a cosine per row) within 1e-6 like the constraint tensor"""
    N, C, H, W, levels = 2, 24, 6, 10, 6
    rng = np.random.default_rng(70 + imp_kernel)
    g = rng.standard_normal((N, C, H, W)).astype(np.float32)
    imp = (np.floor(rng.random((N, 1, H, W)) * (levels + 1)) / levels).astype(np.float32)       # 0 .. 1 in steps of 1/levels
    sc = rng.standard_normal((N, 1, H)).astype(np.float32)                                       # both signs
    op = lic.ImpMapOp(levels, 0.13, 0.7, 0.5, 0.61, 0.61, imp_kernel, 1, 0, False)
    dd, di = op.backward(dev(g), dev(imp), dev(sc))
    alpha = host(op._alpha)
    assert np.allclose(alpha, orc.imp_map_alpha(H, 0.13, 0.61), rtol=1e-6, atol=1e-6)
    rd, ri = orc.imp_map_backward(g, imp, sc, alpha, levels, imp_kernel, np.float32(0.7))
    assert np.array_equal(host(dd), rd) and np.array_equal(host(di), ri)
    # the data gradient is the forward mask applied to the gradient wherever imp*levels is not within 1e-5 below an integer
    fwd_mask = orc.imp_map(np.ones_like(g), imp, levels)[0]
    assert np.array_equal(rd, g * fwd_mask)


@pytest.mark.parametrize("ntop", [1, 2])
def test_quant_training_path(lic, ntop):
    """QuantOp with train=True: update_weight every check_iters calls (bit-exact vs the oracle: same exp / log), then backward --
    data gradient bit-exact (straight through, plus the index-output term when ntop = 2), weight gradient within 1e-5 (the oracle
    sums in double, the kernel per workgroup in a fixed tree; the reference itself uses float atomics in no fixed order)"""
    N, C, H, W, L = 2, 6, 8, 12, 8
    rng = np.random.default_rng(90 + ntop)
    wb = np.concatenate([rng.uniform(-1, 0, (C, 1)), rng.uniform(-2, -0.5, (C, L - 1))], 1).astype(np.float32)
    x = rng.uniform(-1.5, 2.5, (N, C, H, W)).astype(np.float32)
    x[:, 0] = -5.0                                                          # channel 0: everything in level 0 -> trailing levels empty
    x[:, 1] = 50.0                                                          # channel 1: everything in the last level -> first level empty
    op = lic.QuantOp(C, L, 0.9, 2, ntop, 0.1, 0, False)
    wd, cnt = dev(wb), dev(np.zeros((C, L), np.float32))
    w_ref, c_ref = wb.copy(), np.zeros((C, L), np.float32)
    for it in range(3):
        if it == 2:                                                         # iter_ = 2: the update runs before this forward
            w_ref, c_ref = orc.quant_update_weight(w_ref, c_ref, np.float32(0.9))
        out = op.forward(dev(x), wd, cnt, True)
        rt, rq, rc = orc.quant(x, w_ref)
        assert np.array_equal(host(wd), w_ref) and np.array_equal(host(out[0]), rt) and np.array_equal(host(op.count_data_), rc), it
        cnt += op.count_data_ * 0 + dev(rc) * np.float32(-0.001)            # what an optimiser step on `count` does (grad = count_data_)
        c_ref = c_ref + rc * np.float32(-0.001)
        assert np.allclose(host(cnt), c_ref, rtol=0, atol=1e-7)
        c_ref = host(cnt)
    assert not np.array_equal(w_ref, wb)                                    # the update did change the two degenerate channels
    g0 = rng.standard_normal(x.shape).astype(np.float32)
    g1 = rng.standard_normal(x.shape).astype(np.float32)
    tops = [dev(g0), dev(g1)] if ntop == 2 else [dev(g0)]
    dd, wdif, cd = op.backward(tops, dev(x), out[0])
    rd, rw = orc.quant_backward(g0, g1 if ntop == 2 else None, x, rt, rq, w_ref, np.float32(0.1))
    assert np.array_equal(host(dd), rd)
    assert np.allclose(host(wdif), rw, rtol=1e-5, atol=1e-4 * float(np.abs(rw).max()))
    assert cd is op.count_data_


def test_linear_ops_backward_is_the_adjoint(lic):
    """every linear drop-in op with a backward: <forward(x), g> == <x, backward(g)> for random x, g (the defining property of a
    gradient of a linear map) -- pad / trim / cut-edge, pixel shuffle both ways, latitude scaling, context reshape / shift, and the
    data path of ImpMap"""
    rng = np.random.default_rng(77)
    rnd = lambda *s: dev(rng.standard_normal(s).astype(np.float32))

    def adjoint(fwd, bwd, x, gshape):
        y = fwd(x)
        g = rnd(*gshape) if gshape else rnd(*y.shape)
        lhs = float((y.double() * g.double()).sum())
        gx = bwd(g)
        rhs = float((x.double() * gx.double()).sum())
        assert abs(lhs - rhs) <= 1e-4 * (abs(lhs) + 1.0), (lhs, rhs)

    x = rnd(2, 8, 6, 10)
    p = lic.SpherePadOp(2, False, 0, False)
    adjoint(lambda t: p.forward(t)[0].clone(), lambda g: p.backward(g)[0], x, None)
    tr = lic.SphereTrimOp(2, 0, False)
    adjoint(lambda t: tr.forward(t.clone())[0], lambda g: tr.backward(g.clone())[0], rnd(2, 8, 10, 14), None)
    ce = lic.SphereCutEdgeOp(1, 0, False)
    adjoint(lambda t: ce.forward(t)[0].clone(), lambda g: ce.backward(g)[0], x, None)
    for d2w, shape in ((True, (2, 8, 6, 10)), (False, (2, 2, 6, 10))):
        dt = lic.DtowOp(2, d2w, 0, False)
        adjoint(lambda t: dt.forward(t)[0].clone(), lambda g: dt.backward(g)[0], rnd(*shape), None)
    ls = lic.SphereLatScaleOp(3, 0, False)
    wl = rnd(3)
    adjoint(lambda t: ls.forward(t, wl)[0].clone(), lambda g: ls.backward(g, wl)[0], x, None)
    cr = lic.ContextReshapeOp(4, 0, False)
    adjoint(lambda t: cr.forward(t)[0].clone(), lambda g: cr.backward(g)[0], x, None)
    sh = lic.ContexShiftOp(False, 2, 0, False)
    adjoint(lambda t: sh.forward(t)[0].clone(), lambda g: sh.backward(g)[0], x, None)
    levels = 4
    im = lic.ImpMapOp(levels, 0.1, 1.0, 0.5, 0.61, 0.61, 0, 1, 0, False)
    imp = dev((np.floor(rng.random((2, 1, 6, 10)) * (levels + 1)) / levels).astype(np.float32))
    sc = dev(np.ones((2, 1, 6), np.float32))
    adjoint(lambda t: im.forward(t, imp)[0].clone(), lambda g: im.backward(g, imp, sc)[0], x, None)


def test_operator_modules_record_gradients(lic):
    """the nn.Module wrappers of lic360_operator take part in autograd: gradients through them equal those through plain-torch
    formulations of the same maps (pad, cut edge, trim, pixel shuffle, context reshape, latitude scaling), the quantiser is straight
    through with the op's weight / count gradients, the importance map masks the data gradient"""
    import lic360_operator as lo
    import torch.nn.functional as F
    rng = np.random.default_rng(123)
    rnd = lambda *s: dev(rng.standard_normal(s).astype(np.float32))

    def grads(fn, x, g):
        x = x.clone().requires_grad_(True)
        y = fn(x)
        y.backward(g)
        return y.detach().clone(), x.grad.clone()

    x = rnd(2, 8, 6, 10)
    for mod, ref in ((lo.SpherePad(2, 0), lambda t: _sphere_pad_torch(t, 2)),
                     (lo.SphereCutEdge(1, 0), lambda t: t[..., 1:-1, 1:-1]),
                     (lo.Dtow(2, True, 0), lambda t: F.pixel_shuffle(t, 2)),
                     (lo.ContextReshape(4, 0), lambda t: t.reshape(2, 4, 2, 6, 10).permute(0, 1, 3, 4, 2).reshape(-1, 2))):
        y_ref = ref(x)
        g = rnd(*y_ref.shape)
        (y1, g1), (y2, g2) = grads(mod, x, g), grads(ref, x, g)
        assert torch.equal(y1, y2) and torch.allclose(g1, g2, rtol=1e-6, atol=1e-6), type(mod).__name__
    # in-place trim on an activation (x * 1 is a non-leaf)
    tr = lo.SphereTrim(2, 0)
    keep = torch.zeros(6, 10, device="cuda:0")
    keep[2:-2, 2:-2] = 1
    g = rnd(2, 8, 6, 10)
    (y1, g1), (y2, g2) = grads(lambda t: tr(t * 1.0), x, g), grads(lambda t: t * keep, x, g)
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
    # latitude scaling: gradients reach x and the little 1-D net
    ls = lo.SphereLatScaleNet(3, 0).to("cuda:0")
    xs = x.clone().requires_grad_(True)
    ls(xs).backward(g)
    got = [p.grad.clone() for p in ls.net.parameters()]
    ls.zero_grad()
    xr = x.clone().requires_grad_(True)
    wrow = ls.net(ls.data.data).view(3).repeat_interleave(2)                # 6 rows, 3 bands
    (xr * wrow.view(1, 1, 6, 1)).backward(g)
    assert torch.allclose(xs.grad, xr.grad, rtol=1e-6, atol=1e-6)
    for a, p in zip(got, ls.net.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-4, atol=1e-5)
    # quantiser, train mode
    q = lo.QUANT(8, 8, ntop=1, device_id=0).to("cuda:0").train()
    xq = (x * 0.3 + 0.4).clone().requires_grad_(True)
    yq = q(xq)
    yq.backward(g)
    assert torch.equal(xq.grad, g) and q.weight.grad.shape == q.weight.shape and bool(torch.isfinite(q.weight.grad).all())
    assert torch.equal(q.count.grad, q.op[0].count_data_) and float(q.count.grad.sum()) == -float(x.numel())
    # importance map
    im = lo.ImpMap(0.5, 0.1, 1.0, 4, device=0)
    imp = dev(rng.random((2, 1, 6, 10)).astype(np.float32)).requires_grad_(True)
    xi = x.clone().requires_grad_(True)
    out, rt = im(xi, imp)
    out.backward(g)
    mask = (out.detach() != 0) | (xi.detach() == 0)
    assert torch.equal(xi.grad, g * mask) and imp.grad.shape == imp.shape and bool(torch.isfinite(imp.grad).all())


def _viewport_coords_float64(h_out, w_out, theta, phi, fov, H, W):
    """the geometry of ProjectsOp in float64 with numpy only: pinhole rays (x = 1 forward, y right, z up) -> yaw about z by theta*pi, then
    pitch by -phi*pi about the yawed y axis -> longitude / latitude -> ERP pixel coordinates"""
    pi = np.pi
    wf, hf = fov * pi / 2, fov * pi * h_out / w_out / 2
    ys = (np.arange(w_out) - (w_out - 1) / 2) * (2 * np.tan(wf) / (w_out - 1))
    zs = (np.arange(h_out) - (h_out - 1) / 2) * (2 * np.tan(hf) / (h_out - 1))
    yy, zz = np.meshgrid(ys, zs)
    rays = np.stack([np.ones_like(yy), yy, -zz], -1)
    rays /= np.linalg.norm(rays, axis=-1, keepdims=True)

    def rot(axis, ang):
        a = np.asarray(axis, np.float64)
        n = np.linalg.norm(a)
        if n == 0 or ang == 0:
            return np.eye(3)
        k = a / n
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    out = np.empty((14, h_out * w_out, 2))
    for v in range(14):
        yaw = rot([0, 0, 1], theta[v] * pi)
        R = rot(yaw[:, 1], -phi[v] * pi) @ yaw
        d = rays.reshape(-1, 3) @ R.T
        lon, lat = np.arctan2(d[:, 1], d[:, 0]), np.arcsin(np.clip(d[:, 2], -1, 1))
        out[v, :, 0] = lon / pi * (W - 1) / 2 + (W - 1) / 2
        out[v, :, 1] = -2 * lat / pi * (H - 1) / 2 + (H - 1) / 2
    return out


@pytest.mark.parametrize("near", [False, True])
def test_projects_op_viewports(lic, near):
    """ProjectsOp / MultiProject: sampling coordinates == oracle (both fp32 on the host) and == a float64 numpy statement of the geometry
    to 1e-3 pixel (away from the +-pi seam); forward bit-exact vs the oracle; backward (float atomics) within 1e-5; <F x, g> = <x, F^T g>"""
    import lic360_operator as lo
    N, Cc, H, W, ho, wo, fov = 2, 3, 32, 64, 9, 13, 0.5
    rng = np.random.default_rng(5)
    th, ph = list(lo.MultiProject.THETAS), list(lo.MultiProject.PHIS)
    op = lic.ProjectsOp(ho, wo, th, ph, fov, near, 0, False)
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    y = host(op.forward(dev(x))[0])
    tf = host(op._tf)
    tf_ref = orc.projects_tf(ho, wo, np.array(th, np.float32), np.array(ph, np.float32), np.float32(fov), H, W)
    assert np.allclose(tf, tf_ref, rtol=0, atol=2e-4)
    t64 = _viewport_coords_float64(ho, wo, th, ph, fov, H, W)
    seam = np.abs(np.abs(t64[..., 0] - (W - 1) / 2) - (W - 1) / 2) < 0.05                   # lon = +-pi: either end of the row is right
    lat = (0.5 - t64[..., 1] / (H - 1)) * np.pi
    polar = np.cos(lat) < 0.05                                                              # longitude is ill-conditioned next to a pole
    assert np.abs(tf - t64)[..., 1].max() < 1e-3 and np.abs(tf - t64)[..., 0][~seam & ~polar].max() < 1e-3
    assert tf[..., 0].min() >= 0 and tf[..., 0].max() <= W - 1 and tf[..., 1].min() >= 0 and tf[..., 1].max() <= H - 1
    assert abs(tf[1, (ho // 2) * wo + wo // 2, 0] - (W - 1) / 2) < 1e-3 and abs(tf[1, (ho // 2) * wo + wo // 2, 1] - (H - 1) / 2) < 1e-3   # viewport 1 looks at the centre
    assert y.shape == (N * 14, Cc, ho, wo) and np.array_equal(y, orc.projects_forward(x, tf, ho, wo, near))
    g = rng.standard_normal(y.shape).astype(np.float32)
    gd, cnt = op.backward(dev(g))
    rd, rc = orc.projects_backward(g, tf, N, Cc, H, W, near)
    assert np.allclose(host(gd), rd, rtol=1e-5, atol=1e-5) and np.allclose(host(cnt), rc, rtol=1e-5, atol=1e-5)
    lhs, rhs = float((y.astype(np.float64) * g).sum()), float((x.astype(np.float64) * host(gd)).sum())
    assert abs(lhs - rhs) <= 1e-4 * (abs(lhs) + 1)
    # the module: same output, gradient through autograd
    mp = lo.MultiProject(ho, wo, fov, near, 0)
    xt = dev(x).requires_grad_(True)
    out = mp(xt)
    assert np.array_equal(host(out.detach()), y)
    out.backward(dev(g))
    assert np.allclose(host(xt.grad), rd, rtol=1e-5, atol=1e-5)
    # a constant image projects to the same constant (the four bilinear weights sum to one)
    assert np.allclose(host(op.forward(dev(np.full((1, 1, H, W), 0.75, np.float32)))[0]), 0.75, atol=1e-6)


def test_cpp_op_craster_projection(lic):
    """CppOp == oracle bit for bit; the window is centred, full width at the equator and narrow at the poles; a constant image stays
    constant inside it; the mask marks it"""
    N, Cc, H, W = 2, 3, 16, 32
    rng = np.random.default_rng(9)
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    op = lic.CppOp(True, 0, False)
    out, mask = [host(t) for t in op.forward(dev(x))]
    ro, rm = orc.cpp_forward(x, True)
    assert np.array_equal(out, ro) and np.array_equal(mask, rm)
    width = mask[0, 0].sum(1)
    assert width[H // 2] >= W - 2 and width[0] < W / 2 and np.array_equal(width, width[::-1]) and (np.diff(width[:H // 2]) >= 0).all()
    first = mask[0, 0].argmax(1)
    assert (np.abs(first - (W - width - first)) <= 1).all()                    # centred to within a pixel
    c = host(lic.CppOp(False, 0, False).forward(dev(np.full((1, 1, H, W), 0.5, np.float32)))[0])
    assert np.allclose(c[mask[:1, :1] == 1], 0.5, atol=1e-6) and (c[mask[:1, :1] == 0] == 0).all()
    with pytest.raises(lic.Lic360Error):
        lic.CppOp(False, 0, False).forward(dev(np.zeros((1, 1, 8, 8), np.float32)))


def test_viewport_op(lic):
    """ViewportOp vs the oracle (device libm vs host libm: 1e-5 on angles and rays, 1e-4 on the sampled view), and the geometry itself:
    the centre of the viewport looks at (theta, phi); get_viewport_xy of the current direction is the viewport centre; a direction a
    little to the right / up lands right / up of it; a constant image stays constant"""
    N, Cc, H, W, ho, wo, fov = 3, 2, 32, 64, 10, 14, 80.0
    rng = np.random.default_rng(17)
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    tp = np.array([[0.0, 0.0], [1.1, 0.4], [-2.3, -0.7]], np.float32)
    op = lic.ViewportOp(fov, ho, wo, 0, False)
    view, rays0, rota, rays, ang = [host(t) for t in op.forward(dev(x), dev(tp))]
    rv, r0, rr, ry, ra = orc.viewport_forward(x, tp, ho, wo, np.float32(fov))
    assert np.allclose(rays0, r0, atol=1e-6) and np.allclose(rota, rr, atol=1e-6) and np.allclose(rays, ry, atol=1e-5)
    seam = np.abs(np.abs(ra[..., 0]) - np.pi) < 1e-3
    assert np.allclose(ang[..., 1], ra[..., 1], atol=1e-5) and np.allclose(ang[..., 0][~seam], ra[..., 0][~seam], atol=1e-5)
    assert np.allclose(view, rv, rtol=1e-4, atol=1e-4)
    assert np.array_equal(host(op.cal_rota_matrix(dev(tp))), rota) and op.backward(None, None) == []
    # geometry: the mean of the four centre pixels' rays points at (theta, phi)
    ctr = rays[:, ho // 2 - 1:ho // 2 + 1, wo // 2 - 1:wo // 2 + 1].reshape(N, -1, 3).mean(1)
    ctr /= np.linalg.norm(ctr, axis=1, keepdims=True)
    want = np.stack([np.cos(tp[:, 0]) * np.cos(tp[:, 1]), np.sin(tp[:, 0]) * np.cos(tp[:, 1]), np.sin(tp[:, 1])], 1)
    assert np.allclose(ctr, want, atol=1e-5)
    xy = host(op.get_viewport_xy(dev(tp))[0])
    assert np.allclose(xy, orc.viewport_xy(tp, rota, ho, wo, np.float32(fov)), atol=1e-4)
    assert np.allclose(xy, [[wo / 2 - 0.5, ho / 2 - 0.5]] * N, atol=1e-3)
    right_up = host(op.get_viewport_xy(dev(tp + np.array([[0.05, 0.05]], np.float32)))[0])
    assert (right_up[:, 0] > xy[:, 0]).all() and (right_up[:, 1] < xy[:, 1]).all()
    c = host(op.forward(dev(np.full((N, 1, H, W), -1.25, np.float32)), dev(tp))[0])
    assert np.allclose(c, -1.25, atol=1e-6)


def test_context_layouts(lic):
    rng = np.random.default_rng(23)
    x = rng.standard_normal((2, 12, 5, 7)).astype(np.float32)
    cr = lic.ContextReshapeOp(4, 0, False)
    got = host(cr.forward(dev(x))[0])
    assert np.array_equal(got, orc.context_reshape(x, 4))
    assert np.array_equal(got, x.reshape(2, 4, 3, 5, 7).transpose(0, 1, 3, 4, 2).reshape(-1, 3))
    assert np.array_equal(host(cr.backward(dev(got))[0]), x)
    sk = host(lic.ContexShiftOp(False, 3, 0, False).forward(dev(x))[0])
    assert np.array_equal(sk, orc.contex_shift(x, 3, False))
    assert np.array_equal(host(lic.ContexShiftOp(True, 3, 0, False).forward(dev(sk))[0]), x)


def test_conv_ops_follow_weight_data_writes(lic):
    """`param.data.copy_()` does not bump `param._version` (the reference's own QUANT init writes that way): the packed-weight
    cache must not serve stale weights -- encode order repacks per call, decode order at the start of every sweep."""
    rng = np.random.default_rng(3)
    G, cin, cout, H, W = 6, 4, 4, 6, 9
    C, nout = G * cin, G * cout
    w1, b, a = conv_params(rng, 3, nout, C, act=True)
    w2 = (w1 * np.float32(0.5) + np.float32(0.01)).astype(np.float32)
    x = rng.standard_normal((3, C, H, W)).astype(np.float32)
    wt = torch.nn.Parameter(dev(w1), requires_grad=False)
    xd, bd, ad = dev(x), dev(b), dev(a)
    op = lic.CconvEcOp(C, G, nout, 5, 6, 0, False)
    assert np.array_equal(host(op.forward_act_batch(xd, wt, bd, ad)[0]), orc.cconv_ec(x, w1, b, a, G, 6))
    v = wt._version
    wt.data.copy_(dev(w2))
    assert wt._version == v                                          # the trap this test is about
    assert np.array_equal(host(op.forward_act_batch(xd, wt, bd, ad)[0]), orc.cconv_ec(x, w2, b, a, G, 6))
    # decode order: a new sweep (restart) sees the new weights
    ctx = lic.CodeContexOp(0, False)
    p1, p2 = ctx.forward(xd)
    dc = lic.CconvDcOp(C, G, nout, 5, 6, 0, False)
    dc.set_param(p1, p2)
    for wnp in (w1, w2):
        wt.data.copy_(dev(wnp))
        dc.restart()
        out = None
        for p in range(H + W + G - 2):
            out = dc.forward_act_batch(xd, wt, bd, ad)[0]
        assert np.array_equal(host(out), orc.cconv_ec(x, wnp, b, a, G, 6))


def test_operator_extras_plain_torch(lic):
    """GDN / DropGrad / SSIM of lic360_operator (plain torch, outside the accelerated path): formula-level checks."""
    import lic360_operator as lo
    torch.manual_seed(0)
    x = torch.randn(2, 6, 5, 7, device="cuda:0")
    for inverse in (False, True):
        g = lo.GDN(6, 0, inverse)
        with torch.no_grad():
            g.gamma.add_(0.05 * torch.rand_like(g.gamma))
        ped = (2.0 ** -18) ** 2
        beta = g.beta.clamp_min((1e-6 + ped) ** 0.5) ** 2 - ped
        gamma = g.gamma.clamp_min(2.0 ** -18) ** 2 - ped
        norm = torch.sqrt(torch.einsum("ij,njhw->nihw", gamma, x * x) + beta[None, :, None, None])
        want = x * norm if inverse else x / norm
        assert torch.allclose(g(x), want, rtol=1e-5, atol=1e-6)
        assert set(dict(g.named_parameters())) == {"beta", "gamma"}
    a = torch.rand(1, 3, 32, 32, device="cuda:0")
    assert abs(float(lo.SSIM()(a, a)) - 1.0) < 1e-5 and float(lo.SSIM()(a, 1 - a)) < 0.5
    y = torch.ones(3, device="cuda:0", requires_grad=True)
    lo.DropGrad(True)(y).sum().backward()
    assert float(y.grad.abs().sum()) == 0.0
    # the demo's viewport-metric stage: 14 viewports of 171 x 256 per image, PSNR / SSIM of identical inputs
    pr = lo.MultiProject(171, int(171 * 1.5), 0.5, False, 0)
    img = torch.rand((1, 3, 512, 1024), device="cuda:0")
    v = pr(img)
    assert tuple(v.shape) == (14, 3, 171, 256) and abs(float(lo.SSIM(11, 3)(v, v.clone())) - 1.0) < 1e-5


# ------------------------------------------------------------------ encode-order conv on 16x16x4 MFMAs, direct C-ABI call
@pytest.mark.parametrize("case", [(6, 1, 4, False, True, 3, 3, 7, 70), (6, 4, 4, True, True, 3, 3, 5, 66), (48, 4, 4, True, True, 3, 3, 3, 64),
                                  (48, 4, 3, True, False, 3, 6, 4, 20), (48, 1, 4, False, True, 3, 3, 4, 130), (9, 4, 4, True, True, 1, 2, 64, 9),
                                  (7, 1, 4, False, False, 1, 1, 33, 40), (5, 4, 4, False, True, 1, 1, 9, 17), (4, 4, 3, True, False, 1, 2, 130, 8),
                                  (48, 4, 4, True, True, 3, 24, 16, 32), (1, 4, 2, True, True, 1, 1, 5, 5), (3, 1, 1, False, False, 1, 1, 1, 1)],
                         ids=lambda c: "g%d_%dto%d_%s_%dx%d" % (c[0], c[1], c[2], "h" if c[3] else "f", c[7], c[8]))
def test_cconv16_ec_bit_exact(lic, case):
    """lic360_cconv16_ec (v_mfma_f32_16x16x4_f32, K = 4 consecutive input groups) == oracle, on the zero-haloed layout; covers
    both cin, cout < 4, group counts that are not multiples of 4, ragged tiles, many samples per XCD, first layers with cin = 4"""
    import ctypes as C
    G, cin, cout, hidden, act, nb, N, H, W = case
    rng = case_rng(case)
    Cc, nout = G * cin, G * cout
    w, b, a = conv_params(rng, nb if nb > 1 else None, nout, Cc, act=act)
    if nb == 1:
        w, b = w[None], b[None]
        a = None if a is None else a[None]
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    x[rng.random(x.shape) < 0.2] = 0.0
    res = rng.standard_normal((N, nout, H, W)).astype(np.float32)
    constrain = 6 if hidden else 5
    ref = orc.cconv_ec(x, w, b, a, G, constrain) + res
    L = lic._lib
    hp, wp = C.c_int(), C.c_int()
    assert L.lic360_ec16_layout(H, W, C.byref(hp), C.byref(wp)) == 0
    hp, wp = hp.value, wp.value
    pad = lambda t: np.pad(t, ((0, 0), (0, 0), (2, hp - H - 2), (2, wp - W - 2)))
    plan = C.c_void_p(0)
    assert L.lic360_conv_plan_create(Cc, G, nout, 5, constrain, C.byref(plan)) == 0
    assert L.lic360_conv16_supported(plan) == 1
    packed = torch.empty(nb * L.lic360_conv16_packed_floats(plan), dtype=torch.float32, device="cuda:0")
    xd, rd, wd, bd = dev(pad(x)), dev(pad(res)), dev(w), dev(b)
    ad = dev(a) if act else None
    out = torch.zeros((N, nout, hp, wp), dtype=torch.float32, device="cuda:0")
    ctr = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    s = lic._stream(0)
    P = lic._p
    assert L.lic360_conv16_pack(s, plan, P(wd), nb, P(packed)) == 0
    assert L.lic360_cconv16_ec(s, plan, P(xd), P(packed), P(bd), P(ad), P(rd), P(out), N, H, W, nb, N, P(ctr)) == 0, L.lic360_last_error()
    got = host(out)
    L.lic360_conv_plan_destroy(plan)
    assert np.array_equal(got, pad(ref)), "max abs diff %g" % np.abs(got - pad(ref)).max()


# ------------------------------------------------------------------ importance-map net layers (one group, C = 144) on 16x16x4 MFMAs
def _i144_setup(lic, rng, nout, act):
    import ctypes as C
    w, b, a = conv_params(rng, None, nout, 144, act=act)
    L = lic._lib
    plan = C.c_void_p(0)
    assert L.lic360_conv_plan_create(144, 1, nout, 5, 6, C.byref(plan)) == 0
    assert L.lic360_conv144_supported(plan) == 1
    packed = torch.empty(L.lic360_conv144_packed_floats(plan), dtype=torch.float32, device="cuda:0")
    wd = dev(w)
    assert L.lic360_conv144_pack(lic._stream(0), plan, lic._p(wd), lic._p(packed)) == 0, L.lic360_last_error()
    return w, b, a, plan, packed


@pytest.mark.parametrize("case", [(2, 7, 21, 144, True, 2), (1, 4, 6, 49, False, 0), (3, 32, 64, 144, True, 2), (1, 1, 1, 20, True, 0), (1, 33, 17, 49, False, 0)],
                         ids=lambda c: "n%d_%dx%d_to%d" % c[:4])
def test_cconv144_ec_bit_exact(lic, case):
    """lic360_cconv144_ec (hidden / last layers of the importance-map net) == oracle; output into the haloed layout or plain NCHW"""
    import ctypes as C
    N, H, W, nout, act, ooff = case
    rng = case_rng(case)
    w, b, a, plan, packed = _i144_setup(lic, rng, nout, act)
    x = rng.standard_normal((N, 144, H, W)).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = 0.0
    res = rng.standard_normal((N, nout, H, W)).astype(np.float32)
    ref = orc.cconv_ec(x, w, b, a, 1, 6) + res
    L = lic._lib
    hp, wp = C.c_int(), C.c_int()
    assert L.lic360_ec144_layout(H, W, C.byref(hp), C.byref(wp)) == 0
    hp, wp = hp.value, wp.value
    pad = lambda t: np.pad(t, ((0, 0), (0, 0), (2, hp - H - 2), (2, wp - W - 2)))
    xd, bd = dev(pad(x)), dev(b)
    ad = dev(a) if act else None
    if ooff:
        rd, out, oplane, opitch = dev(pad(res)), torch.zeros((N, nout, hp, wp), dtype=torch.float32, device="cuda:0"), hp * wp, wp
    else:
        rd, out, oplane, opitch = dev(res), torch.zeros((N, nout, H, W), dtype=torch.float32, device="cuda:0"), H * W, W
    P = lic._p
    assert L.lic360_cconv144_ec(lic._stream(0), plan, P(xd), P(packed), P(bd), P(ad), P(rd), P(out), N, H, W, oplane, opitch, ooff) == 0, L.lic360_last_error()
    got = host(out)
    L.lic360_conv_plan_destroy(plan)
    want = pad(ref) if ooff else ref
    assert np.array_equal(got, want), "max abs diff %g" % np.abs(got - want).max()


@pytest.mark.parametrize("case", [(2, 6, 9, 144, True), (1, 32, 64, 49, False), (3, 5, 3, 144, True), (2, 32, 7, 144, True),
                                  (1, 64, 20, 144, True), (2, 45, 70, 49, False)], ids=lambda c: "n%d_%dx%d_to%d" % c[:4])   # the last two: diagonals longer than one task window
def test_cconv144_dc_planes_bit_exact(lic, case):
    """decode order on the zero-padded diagonal-major layout: after every plane the persistent output equals the oracle's"""
    import ctypes as C
    N, H, W, nout, act = case
    rng = case_rng(case)
    w, b, a, plan, packed = _i144_setup(lic, rng, nout, act)
    x = rng.standard_normal((N, 144, H, W)).astype(np.float32)
    res = rng.standard_normal((N, nout, H, W)).astype(np.float32)
    L = lic._lib
    rows, pitch = C.c_int(), C.c_int()
    assert L.lic360_dc144_layout(H, W, C.byref(rows), C.byref(pitch)) == 0
    rows, pitch = rows.value, pitch.value
    th, tw = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")

    def skew(t):
        o = np.zeros(t.shape[:2] + (rows, pitch), np.float32)
        o[:, :, th + tw + 4, th + 2] = t
        return o
    xd, rd, bd = dev(skew(x)), dev(skew(res)), dev(b)
    ad = dev(a) if act else None
    out = torch.zeros((N, nout, rows, pitch), dtype=torch.float32, device="cuda:0")
    idx, pidx = orc.code_contex(H, W)
    ref = np.zeros((N, nout, H, W), np.float32)
    P = lic._p
    for s in range(H + W - 1):
        orc.cconv_dc_plane(x, w, b, a, ref, 1, 6, idx, pidx, s)
        assert L.lic360_cconv144_dc_plane(lic._stream(0), plan, P(xd), P(packed), P(bd), P(ad), P(rd), P(out), N, H, W, s) == 0, L.lic360_last_error()
        if s in (0, 3, H - 1, H + W - 2):
            on = (th + tw <= s)
            assert np.array_equal(host(out), skew((ref + res) * on)), "plane %d" % s
    L.lic360_conv_plan_destroy(plan)
    assert np.array_equal(ref, orc.cconv_ec(x, w, b, a, 1, 6))


@pytest.mark.parametrize("G,B,H,W", [(6, 2, 6, 22), (48, 1, 8, 16), (3, 1, 4, 6), (9, 3, 10, 18)])   # even H, W: util.latent draws the importance level per 2x2 cell
def test_cconv16_last_layer_with_fused_tables(lic, G, B, H, W):
    """lic360_cconv16_ec_tables (N1: last conv layer + softmax / sigma floor / erf CDF / fix-up / record write in one kernel) ==
    oracle last layer -> per-plane TileExtractBatch -> EntropyBatchGmmTable -> (T[sym], T[sym+1]) in coding order"""
    import ctypes as C
    rng = np.random.default_rng(G * 100 + H)
    Cc, nout = G * 4, G * 3
    w, b, _ = conv_params(rng, 3, nout, Cc, act=False)
    b[1] += 2.0
    x = (0.5 * rng.standard_normal((3 * B, Cc, H, W))).astype(np.float32)
    code, mask = [], []
    for i in range(B):
        c, m, _ = latent(np.random.default_rng(i), G, H, W)
        code.append(c)
        mask.append(m)
    code, mask = np.concatenate(code, 0), np.concatenate(mask, 0)
    y = orc.cconv_ec(x, w, b, None, G, 6)                                    # [3B, 3G, H, W], net-major
    idx, pidx = orc.code_contex(H, W)
    P = H + W + G - 2
    want = np.zeros((B, G * H * W, 2), np.uint32)
    plane_start = np.zeros(P + 1, np.int32)
    k = 0
    z, lab, mk = np.zeros(3 * 3 * H * W, np.float32), np.zeros(H * W, np.float32), np.zeros(H * W, np.float32)
    for p in range(P):
        plane_start[p] = k
        tn = 0
        for bi in range(B):
            yb = np.ascontiguousarray(y[bi::B])                              # the three nets of image bi
            tn = orc.tile_extract_batch(yb, z, G, idx, pidx, p)
            tab = orc.gmm_table_batch(z, 3 * H * W, tn).astype(np.int64)
            orc.tile_extract(code[bi:bi + 1], lab, G, True, idx, pidx, p)
            orc.tile_extract(mask[bi:bi + 1], mk, G, True, idx, pidx, p)
            s = lab[:tn].astype(np.int64)
            keep = mk[:tn] >= 0.5
            want[bi, k:k + tn, 0] = np.where(keep, tab[np.arange(tn), s], 0)
            want[bi, k:k + tn, 1] = np.where(keep, tab[np.arange(tn), s + 1], 0)
        k += tn
    plane_start[P] = k
    assert k == G * H * W
    L = lic._lib
    hp, wp = C.c_int(), C.c_int()
    assert L.lic360_ec16_layout(H, W, C.byref(hp), C.byref(wp)) == 0
    pad = lambda t: np.pad(t, ((0, 0), (0, 0), (2, hp.value - H - 2), (2, wp.value - W - 2)))
    plan = C.c_void_p(0)
    assert L.lic360_conv_plan_create(Cc, G, nout, 5, 6, C.byref(plan)) == 0
    packed = torch.empty(3 * L.lic360_conv16_packed_floats(plan), dtype=torch.float32, device="cuda:0")
    xd, wd, bd, cd, md = dev(pad(x)), dev(w), dev(b), dev(code), dev(mask)
    pd, psd = dev(pidx.astype(np.int32)), dev(plane_start)
    rec = torch.full((B, G * H * W, 2), -1, dtype=torch.int32, device="cuda:0")
    ctr = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    s, Pp = lic._stream(0), lic._p
    assert L.lic360_conv16_pack_tables(s, plan, Pp(wd), 3, Pp(packed)) == 0
    assert L.lic360_cconv16_ec_tables(s, plan, Pp(xd), Pp(packed), Pp(bd), Pp(cd), Pp(md), Pp(pd), Pp(psd), Pp(rec), B, H, W, Pp(ctr)) == 0, L.lic360_last_error()
    got = host(rec).astype(np.int64).astype(np.uint32)
    L.lic360_conv_plan_destroy(plan)
    bad = np.argwhere((got != want).any(-1))
    assert len(bad) == 0, "%d of %d records differ, first (image, k): %s got %s want %s" % (
        len(bad), B * G * H * W, bad[:4].tolist(), got[tuple(bad[0])].tolist(), want[tuple(bad[0])].tolist())


# ------------------------------------------------------------------ decode-order conv of the latent nets, plane by plane, direct C-ABI call
@pytest.mark.parametrize("case", [(12, 4, True, True, 2, 48, 40, 12), (12, 4, True, True, 1, 48, 64, 20), (6, 3, True, False, 3, 144, 18, 7),
                                  (15, 4, False, True, 2, 32, 30, 30), (8, 4, True, True, 3, 3, 6, 9), (48, 4, True, True, 3, 48, 8, 16),
                                  (48, 3, True, False, 3, 6, 64, 20), (4, 4, False, True, 1, 1, 33, 40), (4, 3, True, False, 2, 96, 30, 6),
                                  (20, 2, True, True, 2, 34, 17, 3), (48, 4, True, True, 3, 48, 64, 7), (16, 4, True, True, 1, 17, 50, 30),
                                  (8, 4, False, True, 3, 144, 33, 21), (12, 4, True, True, 1, 40, 50, 12), (12, 4, True, True, 2, 64, 33, 9),
                                  (9, 4, True, True, 1, 72, 64, 30), (12, 3, True, False, 1, 144, 61, 23), (48, 4, False, True, 3, 48, 64, 20, 1),
                                  (12, 4, False, True, 1, 40, 50, 12, 1), (8, 4, False, True, 3, 3, 20, 9, 1)],
                         ids=lambda c: "g%d_%dto%d_%s_n%d_%dx%d" % (c[0], c[8] if len(c) > 8 else 4, c[1], "h" if c[2] else "f", c[5], c[6], c[7]))
def test_cconv4_dc_planes_packed_bit_exact(lic, case):
    """the production decode kernel (4x4x1 MFMAs, lic360_cconv4_dc_plane) plane by plane with its sample packing -- TAPE packing (round 5): the
    row windows of 2..6 consecutive samples of an XCD's list laid end to end over the lanes of as few tasks as fit, windows cut between tasks
    where they do not (tests/test_dc_tape.py checks the host side's rules on the CPU); blocks of full-length diagonals keep one sample per task
    inside the same launch.  Cases: tapes of 6 (48 / 144 samples per net), 2 (16), 5 (40), 4 (32), 3 (72: nine per list) samples; short, 64-row
    and 61-row (odd, cut) diagonals; every case has more than 128 three-group tasks per launch (fewer switch the kernel to its
    one-group-per-task latency mode, which does not pack); from the fifth case on also few samples (latency mode), odd sample counts,
    cout = 2 / 3, first-layer constraint, widths below 5, 144 samples; the last three: the FIRST layer's instantiation (cin = 1)"""
    _cconv4_dc_planes(lic, case, "lic360_cconv4_dc_plane", ("lic360_conv4_supported", "lic360_conv4_packed_floats", "lic360_conv4_pack"))


def _cconv4_dc_planes(lic, case, entry, packfns):
    """one decode-order layer of the latent nets on the lic360_dc4_layout, plane by plane: after every checked plane the persistent output
    equals the oracle's (extension/cconv_dc_cuda.cu:313-398) + residual"""
    import ctypes as C
    G, cout, hidden, act, nb, N, H, W = case[:8]
    cin = case[8] if len(case) > 8 else 4
    rng = np.random.default_rng(1000 + 7 * G + 131 * N + 17 * H + W)
    Cc, nout = G * cin, G * cout
    w, b, a = conv_params(rng, nb if nb > 1 else None, nout, Cc, act=act)
    if nb == 1:
        w, b = w[None], b[None]
        a = None if a is None else a[None]
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    x[rng.random(x.shape) < 0.2] = 0.0
    res = rng.standard_normal((N, nout, H, W)).astype(np.float32)
    constrain = 6 if hidden else 5
    L = lic._lib
    rows, pitch, row0, col0 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.lic360_dc4_layout(H, W, C.byref(rows), C.byref(pitch), C.byref(row0), C.byref(col0)) == 0
    rows, pitch, row0, col0 = rows.value, pitch.value, row0.value, col0.value
    th, tw = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")

    def skew(t, fill=0.0):
        o = np.full(t.shape[:2] + (rows, pitch), fill, np.float32)
        o[:, :, th + tw + row0, th + col0] = t
        return o

    def to_dev(t):                                                         # planes + the slack the band fetches may touch
        buf = torch.zeros(L.lic360_conv4_buffer_floats(0, t.shape[0] * t.shape[1], H, W), dtype=torch.float32, device="cuda:0")
        buf[:t.size] = torch.from_numpy(t.reshape(-1)).to("cuda:0")
        return buf
    plan = C.c_void_p(0)
    assert L.lic360_conv_plan_create(Cc, G, nout, 5, constrain, C.byref(plan)) == 0
    assert getattr(L, packfns[0])(plan) == 1
    getattr(L, packfns[1]).restype = C.c_long
    getattr(L, packfns[1]).argtypes = [C.c_void_p]
    packed = torch.empty(nb * getattr(L, packfns[1])(plan), dtype=torch.float32, device="cuda:0")
    wd, bd = dev(w), dev(b)
    ad = dev(a) if act else None
    xd, rd = to_dev(skew(x)), to_dev(skew(res))
    out = to_dev(skew(np.zeros_like(res)))
    s = lic._stream(0)
    P = lic._p
    assert getattr(L, packfns[2])(s, plan, P(wd), nb, P(packed)) == 0, L.lic360_last_error()
    idx, pidx = orc.code_contex(H, W)
    ref = np.zeros((N, nout, H, W), np.float32)
    nplanes = H + W + G - 2
    check = {0, 1, 2, G - 1, G, H - 1, H + 1, W, nplanes // 2, nplanes - 2, nplanes - 1}
    # x is complete from the start (the encoder's view of the same data): the causal rule decides what a plane may read
    for p in range(nplanes):
        orc.cconv_dc_plane(x, w, b, a, ref, G, constrain, idx, pidx, p)
        assert getattr(L, entry)(s, plan, P(xd), P(packed), P(bd), P(ad), P(rd), P(out), N, H, W, nb, p, N) == 0, L.lic360_last_error()
        if p in check:
            got = host(out)[:N * nout * rows * pitch].reshape(N, nout, rows, pitch)
            g = np.arange(G).repeat(cout)[None, :, None, None]
            on = (th + tw)[None, None] + g <= p                           # outputs of planes <= p
            want = skew(np.where(on, ref + res, 0.0).astype(np.float32))
            assert np.array_equal(got, want), "plane %d: %d cells differ, max abs diff %g" % (p, int((got != want).sum()), np.abs(got - want).max())
    L.lic360_conv_plan_destroy(plan)
    assert np.array_equal(ref, orc.cconv_ec(x, w, b, a, G, constrain))


# ------------------------------------------------------------------ streaming ops at the shapes the bench times
@pytest.mark.parametrize("shape", [(2, 192, 260, 516), (2, 192, 132, 260), (1, 5, 9, 14), (1, 70000, 6, 7)],
                         ids=lambda s: "x".join(map(str, s)))
def test_sphere_inplace_and_trim_at_bench_shapes(lic, shape):
    """in-place apron refresh and trim on the padded planes of the transforms (260x516, 132x260: 16-byte paths, multi-chunk planes), a
    shape whose padded width is no multiple of 4 (scalar paths) and more than 65535 planes (the fall-back kernels)"""
    rng = np.random.default_rng(31)
    y = rng.standard_normal(shape).astype(np.float32)
    want = orc.sphere_pad_inplace(y.copy(), 2)
    yd = dev(y)
    lic.SpherePadOp(2, True, 0, False).forward(yd)
    assert np.array_equal(host(yd), want)
    lic.SphereTrimOp(2, 0, False).forward(yd)
    assert np.array_equal(host(yd), orc.sphere_trim(want.copy(), 2))
    inner = want[:, :, 2:-2, 2:-2]
    assert np.array_equal(host(lic.SphereCutEdgeOp(2, 0, False).forward(dev(want))[0]), inner)


@pytest.mark.parametrize("shape", [(2, 192, 32, 64), (1, 3, 512, 1024), (1, 66000, 3, 5)], ids=lambda s: "x".join(map(str, s)))
def test_sphere_pad_at_bench_shapes(lic, shape):
    """out-of-place pad: latent planes 32x64 -> 36x68, the first layer's 512x1024 image, > 65535 planes"""
    rng = np.random.default_rng(32)
    x = rng.standard_normal(shape).astype(np.float32)
    assert np.array_equal(host(lic.SpherePadOp(2, False, 0, False).forward(dev(x))[0]), orc.sphere_pad(x, 2))


def test_pointwise_ops_at_bench_shapes(lic):
    """dtow / wtod, imp_map, quant, dquant on [2, 192, 32, 64] (what one encode / decode of two 512x1024 images moves)"""
    rng = np.random.default_rng(33)
    N, C, H, W, levels = 2, 192, 32, 64, 48
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    up = host(lic.DtowOp(2, True, 0, False).forward(dev(x))[0])
    assert np.array_equal(up, orc.dtow(x, 2, True)) and up.shape == (N, 48, 64, 128)
    assert np.array_equal(host(lic.DtowOp(2, False, 0, False).forward(dev(up))[0]), x)
    imp = np.floor(rng.random((N, 1, H, W)) * levels).astype(np.float32) / levels
    out = lic.ImpMapOp(levels, 0.1, 1.0, 0.5, 0.61, 0.61, 0, 3, 0, False).forward(dev(x), dev(imp))
    ro, rm = orc.imp_map(x, imp, levels)
    assert np.array_equal(host(out[0]), ro) and np.array_equal(host(out[2]), rm)
    wb = np.concatenate([rng.uniform(-1, 0, (C, 1)), rng.uniform(-2, -0.5, (C, 7))], 1).astype(np.float32)
    q = lic.QuantOp(C, 8, 0.9, 100, 2, 0.1, 0, False)
    top, qidx = q.forward(dev(x), dev(wb), dev(np.zeros((C, 8), np.float32)), False)
    rt, rq, rc = orc.quant(x, wb)
    assert np.array_equal(host(top), rt) and np.array_equal(host(qidx), rq) and np.array_equal(host(q.count_data_), rc)
    assert np.array_equal(host(lic.DquantOp(C, 8, 0, False).forward(dev(rq), dev(rm), dev(wb))[0]), orc.dquant(rq, rm, wb))
