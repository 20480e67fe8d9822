"""CPU tests pinning the oracle's op restatements by INDEPENDENT numpy / torch-CPU / scipy formulations
(SURVEY.md §8c pin (2)): the reference has no CPU kernels and no golden vectors for these ops."""
import numpy as np
import pytest
import torch
from scipy import special

import oracle as orc
from util import conv_params


def test_code_contex_order():
    H, W = 5, 7
    idx, pidx = orc.code_contex(H, W)
    want = [(s, r, s - r) for s in range(H + W - 1) for r in range(max(0, s - W + 1), min(H - 1, s) + 1)]
    assert [(idx[k] + idx[k + H * W], idx[k], idx[k + H * W]) for k in range(H * W)] == want
    assert pidx.tolist() == [sum(1 for w in want if w[0] < s) for s in range(H + W - 1)] + [H * W]
    # every (g,h,w) exactly once over the G+H+W-2 planes
    G = 4
    assert sum(orc.plane_len(p, G, H, W, pidx) for p in range(H + W + G - 2)) == G * H * W
    assert orc.plane_len(H + W + G - 2, G, H, W, pidx) == 0


@pytest.mark.parametrize("hidden", [False, True])
@pytest.mark.parametrize("G,cin,cout", [(5, 4, 4), (5, 1, 4), (1, 12, 7), (4, 4, 3)])
def test_cconv_matches_masked_conv2d(G, cin, cout, hidden):
    """mask rule of extension/mask_constrain_cuda.cu:17-41 (training-time definition of the same conv)."""
    rng = np.random.default_rng(G * 100 + cin * 10 + cout + hidden)
    C, nout, H, W = G * cin, G * cout, 6, 7
    w, b, a = conv_params(rng, None, nout, C, act=True)
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    o, ti, kh, kw = np.meshgrid(np.arange(nout), np.arange(C), np.arange(5), np.arange(5), indexing="ij")
    keep = (kh + kw + ti // cin <= o // cout + 4) if hidden else (kh + kw + ti // cin < o // cout + 4)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w * keep).double(), torch.from_numpy(b).double(), padding=2)
    ref = torch.where(ref > 0, ref, ref * torch.from_numpy(a).double().view(1, -1, 1, 1)).numpy()
    got = orc.cconv_ec(x, w, b, a, G, 6 if hidden else 5)
    assert np.abs(got - ref).max() < 1e-4


def test_cconv_dc_equals_ec_and_reduction_tree():
    rng = np.random.default_rng(5)
    G, cin, cout, H, W = 4, 4, 4, 5, 6
    w, b, a = conv_params(rng, 3, G * cout, G * cin, act=True)
    x = rng.standard_normal((3, G * cin, H, W)).astype(np.float32)
    ec = orc.cconv_ec(x, w, b, a, G, 6)
    idx, pidx = orc.code_contex(H, W)
    dc = np.zeros_like(ec)
    for p in range(H + W + G - 2):
        orc.cconv_dc_plane(x, w, b, a, dc, G, 6, idx, pidx, p)
    assert np.array_equal(ec, dc)
    # hand-evaluated tree for one output scalar: 100 fmaf chains, then +64, +32, +16, +8, +4, +2, +1
    n, o, th, tw = 1, 9, 2, 3
    g = o // cout
    p = np.zeros(128, np.float32)
    for lane in range(100):
        kw_, kh_, gid = lane % 5, (lane // 5) % 5, lane // 25
        ph, pw = th - 2 + kh_, tw - 2 + kw_
        if not (0 <= ph < H and 0 <= pw < W):
            continue
        nch = min(G * cin, (g + 4 - kh_ - kw_ + 1) * cin)
        s = np.float32(0)
        for ti in range(gid, nch, cin):
            s = np.float32(np.float64(x[n, ti, ph, pw]) * np.float64(w[n, o, ti, kh_, kw_]) + np.float64(s))   # fma: exact product, one rounding
        p[lane] = s
    for off in (64, 32, 16, 8, 4, 2, 1):
        p[:off] = p[:off] + p[off:2 * off]
    v = np.float32(p[0] + b[n, o])
    v = v if v > 0 else np.float32(v * a[n, o])
    assert v == ec[n, o, th, tw]


def test_layout_ops_against_numpy():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 8, 6, 10)).astype(np.float32)
    assert np.array_equal(orc.dtow(x, 2, True), torch.nn.functional.pixel_shuffle(torch.from_numpy(x), 2).numpy())
    assert np.array_equal(orc.dtow(x, 2, False), torch.nn.functional.pixel_unshuffle(torch.from_numpy(x), 2).numpy())
    assert np.array_equal(orc.context_reshape(x, 4), x.reshape(2, 4, 2, 6, 10).transpose(0, 1, 3, 4, 2).reshape(-1, 2))
    sk = orc.contex_shift(x, 2, False)
    for c in (0, 3, 7):
        for h in (0, 5):
            for w in (0, 9):
                assert sk[1, c, h + w + c // 2, w] == x[1, c, h, w]
    assert np.array_equal(orc.contex_shift(sk, 2, True), x)
    pad = 2
    ref = orc.sphere_pad(x, pad)
    mid = np.concatenate([x[..., -pad:], x, x[..., :pad]], -1)
    assert np.array_equal(ref, np.concatenate([mid[:, :, :pad][:, :, ::-1, ::-1], mid, mid[:, :, -pad:][:, :, ::-1, ::-1]], 2))
    y = ref.copy()
    y[:, :, :pad] = 9
    y[..., -pad:] = 9
    assert np.array_equal(orc.sphere_pad_inplace(y, pad), ref)
    t = orc.sphere_trim(ref.copy(), pad)
    assert np.array_equal(t[:, :, pad:-pad, pad:-pad], x) and t.sum() == pytest.approx(x.sum(), rel=1e-5)
    assert np.array_equal(orc.sphere_cut_edge(ref, pad), x)
    wgt = rng.random((1, 1, 3)).astype(np.float32)
    assert np.array_equal(orc.sphere_lat_scale(x, wgt, 3), x * np.repeat(wgt.reshape(-1), 2)[None, None, :, None])


def test_impmap_quant_against_numpy():
    rng = np.random.default_rng(7)
    N, C, H, W, levels = 1, 12, 4, 5, 6
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    L = rng.integers(0, levels + 1, (N, 1, H, W))
    imp = (L / levels).astype(np.float32)
    out, mask = orc.imp_map(x, imp, levels)
    want = (np.arange(C)[None, :, None, None] < L * (C // levels))
    assert np.array_equal(mask, want.astype(np.float32)) and np.array_equal(out, x * want)
    assert np.array_equal(orc.imp2mask(L.astype(np.float32), levels, C), want.astype(np.float32))
    # quantiser: centres c0, c0+e^{w1}, ...; nearest centre
    wb = np.concatenate([rng.uniform(-1, 0, (C, 1)), rng.uniform(-2, -0.5, (C, 7))], 1).astype(np.float32)
    top, qidx, count = orc.quant(x, wb)
    centres = np.cumsum(np.concatenate([wb[:, :1], np.exp(wb[:, 1:].astype(np.float64))], 1), 1)
    near = np.abs(x[0][:, None] - centres[:, :, None, None]).argmin(1)
    assert (near == qidx[0]).mean() > 0.999                      # ties / rounding at bin edges only
    assert np.abs(top[0] - np.take_along_axis(centres[:, :, None, None].repeat(H, 2).repeat(W, 3), qidx[0][:, None].astype(int), 1)[:, 0]).max() < 1e-5
    assert count.sum() == -x.size
    m = (rng.random(x.shape) > 0.5).astype(np.float32)
    dq = orc.dquant(qidx, m, wb)
    assert np.abs(dq[0] - np.where(m[0] > 0, top[0], centres[:, 0][:, None, None])).max() < 1e-5


def test_exact_math_accuracy():
    """lic360_exact_math.h routines through the table / likelihood ops vs float64."""
    rng = np.random.default_rng(8)
    tn = 2000
    w = rng.standard_normal((tn, 3)).astype(np.float32)
    s = rng.uniform(0.05, 3, (tn, 3)).astype(np.float32)
    m = rng.uniform(-4, 4, (tn, 3)).astype(np.float32)
    tab = orc.gmm_table(w.copy(), s.copy(), m.copy(), tn)
    sw = np.exp(w.astype(np.float64) - w.max(1, keepdims=True))
    sw /= sw.sum(1, keepdims=True)
    edges = np.arange(1, 8) - 4.0
    cdf = (sw[:, None, :] * (0.5 + 0.5 * special.erf((edges[None, :, None] - m[:, None, :]) / (s[:, None, :].astype(np.float64) + 1e-6) / np.sqrt(2)))).sum(-1)
    exact = np.floor(65536 * cdf + 0.5)
    well = np.all(np.diff(exact, axis=1) > 2, axis=1) & (exact[:, 0] > 2) & (exact[:, -1] < 65533)
    assert well.sum() > tn // 4 and np.abs(tab[well, 1:8] - exact[well]).max() <= 1
    assert np.all(np.diff(tab, axis=1) > 0) and np.all(tab[:, 0] == 0) and np.all(tab[:, 8] == 65536)
    lg = (rng.standard_normal((50, 49)) * 4).astype(np.float32)
    t2 = orc.entropy_table(lg.reshape(-1), 50, 49)
    pr = np.exp(lg.astype(np.float64) - lg.max(1, keepdims=True))
    pr /= pr.sum(1, keepdims=True)
    assert np.abs(np.diff(t2, axis=1) - pr * 65536).max() < 60       # per-bin rounding + fix-up redistribution
    assert np.all(np.diff(t2, axis=1) > 0) and np.all(t2[:, -1] == 65536)
    M = 3000
    ww = rng.random((M, 3)).astype(np.float32)
    ww /= ww.sum(1, keepdims=True)
    d = rng.uniform(0.1, 3, (M, 3)).astype(np.float32)
    mu = rng.uniform(-3, 3, (M, 3)).astype(np.float32)
    lab = rng.integers(-3, 4, (M, 1)).astype(np.float32)
    loss = orc.entropy_gmm(ww, d, mu, lab)[0]
    phi = lambda z: 0.5 + 0.5 * special.erf(z / np.sqrt(2))
    p = (ww.astype(np.float64) * (phi((lab + 0.5 - mu) / d) - phi((lab - 0.5 - mu) / d))).sum(1)
    big = p > 0.02
    assert np.abs(loss - (-np.log(p + 1e-7)))[big].max() < 1e-5


def test_oracle_codec_roundtrip_small():
    """Oracle pipeline D1/D2/D3 is self-consistent: decode(encode(x)) == x, re-encode is byte-identical."""
    import ref_codec as rc
    from util import latent
    rng = np.random.default_rng(9)
    G, H, W = 4, 6, 8
    code, mask, _ = latent(rng, G, H, W)
    layers = rc.make_main_params(11, G)
    data = rc.encode_main(code, mask, layers, G)
    out = rc.decode_main(data, mask, layers, G)
    assert np.array_equal(out, code * mask)
    assert rc.encode_main(out + (1 - mask) * code, mask, layers, G) == data
    lv = rng.integers(0, 9, (1, 1, 4, 6)).astype(np.float32)
    il = rc.make_imp_params(12, cpg=24, nsym=9)
    assert np.array_equal(rc.decode_imp(rc.encode_imp(lv, il, nsym=9), il, 4, 6, nsym=9), lv)


@pytest.mark.parametrize("ngroup,c_in,c_out,k,constrain", [(6, 4, 4, 5, 6), (6, 1, 4, 5, 5), (1, 8, 5, 5, 6), (4, 3, 2, 3, 5)])
def test_mask_constrain_rule(ngroup, c_in, c_out, k, constrain):
    """orc_mask_constrain against the rule written out with numpy index grids (extension/mask_constrain_cuda.cu:17-41)"""
    rng = np.random.default_rng(k + ngroup)
    w = rng.standard_normal((c_out * ngroup, c_in * ngroup, k, k)).astype(np.float32)
    tn = (np.arange(c_out * ngroup) // c_out)[:, None, None, None]
    tc = (np.arange(c_in * ngroup) // c_in)[None, :, None, None]
    th, tw = np.arange(k)[None, None, :, None], np.arange(k)[None, None, None, :]
    dead = (tw + th + tc >= tn + k - 1) if constrain == 5 else (tw + th + tc > tn + k - 1)
    assert np.array_equal(orc.mask_constrain(w, ngroup, constrain), np.where(dead, np.float32(0), w))
