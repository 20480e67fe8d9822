"""Dead-cone skip of the fused latent codec (csrc/need.h, need_kernels.hip).

The reference evaluates every output of the entropy nets and coder.cpp:79 skips the masked symbols afterwards; the fused codec does not compute the
outputs no coded symbol can observe.  Checked here:
  * the need maps against a brute-force reachability statement of the cone (consumer (g, p) of layer l + 1 reads (g', p + d) of layer l for
    g' <= g - d_y - d_x: cconv_ec_cuda.cu:288-290 with hidden = 1; the residual blocks read (g, p) itself);
  * bitstreams / symbols with the skip on == with it off == the oracle's goldens, on smooth and i.i.d. masks, with every
    interior cell of every activation buffer poisoned beforehand (a huge FINITE value: zero-weight MFMA lanes do read dead cells and
    0 * NaN is NaN -- what the test proves is that no dead value reaches a live cell through a non-zero weight);
  * the task lists themselves: every live cell is stored exactly once, no piece breaks the decode kernel's lane rules.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from util import latent, latent_smooth, make_main_params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POISON = 1.0e10


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def brute_need(mask):
    """mask [G, H, W] bool -> need [12, H, W]: highest live group per layer and position, by explicit reachability on boolean arrays:
    (g', q) of layer l is live iff some live (g, p) of layer l + 1 reads it with a non-zero weight: q = p + d, d in [-2, 2]^2, g' <= g - d_y - d_x"""
    G, H, W = mask.shape
    live = mask.copy()
    out = np.full((12, H, W), -1, np.int64)

    def top(lv):
        return np.where(lv, np.arange(G)[:, None, None], -1).max(0)
    out[11] = top(live)
    for l in range(10, -1, -1):
        nxt = np.zeros_like(live)
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                # consumers p = q - d, seen from the input cell q
                sh = np.zeros_like(live)
                ys, xs = slice(max(0, dy), min(H, H + dy)), slice(max(0, dx), min(W, W + dx))
                yc, xc = slice(max(0, -dy), min(H, H - dy)), slice(max(0, -dx), min(W, W - dx))
                sh[:, ys, xs] = live[:, yc, xc]
                suf = np.logical_or.accumulate(sh[::-1], 0)[::-1]              # suf[g] = some consumer group >= g is live there
                k = dy + dx                                                    # g' is read by consumer groups g >= g' + k
                if k >= 0:
                    nxt[:G - k] |= suf[k:]
                else:
                    nxt[-k:] |= suf[:G + k]
                    nxt[:-k] |= suf[:1]                                        # g' + k < 0: every consumer group reads it
        live = nxt
        out[l] = top(live)
    return out


@pytest.mark.parametrize("G,H,W,kind", [(6, 8, 12, "iid"), (12, 10, 7, "smooth"), (48, 12, 16, "smooth"), (5, 4, 5, "empty")])
def test_need_maps_match_reachability(G, H, W, kind):
    import ctypes as C
    from lic360 import _lib, _chk, _p
    rng = np.random.default_rng(G * 1000 + H)
    if kind == "smooth":
        _, mask, _ = latent_smooth(rng, G, H - H % 2, W - W % 2)
        m = np.zeros((1, G, H, W), np.float32)
        m[:, :, :H - H % 2, :W - W % 2] = mask
        mask = m
    elif kind == "empty":
        mask = np.zeros((1, G, H, W), np.float32)
        mask[0, :2, 1, 2] = 1
    else:
        mask = (rng.random((1, G, H, W)) < 0.3).astype(np.float32)
        mask = np.minimum.accumulate(mask, 1)                                  # a prefix in g, as ImpMap's masks are
    out = torch.zeros((1, 12, H, W), dtype=torch.int8, device="cuda:0")
    _chk(_lib.lic360_need_maps(C.c_void_p(torch.cuda.current_stream().cuda_stream), _p(dev(mask)), 1, G, H, W, _p(out)))
    assert np.array_equal(out.cpu().numpy()[0].astype(np.int64), brute_need(mask[0] > 0.5))


def _codec(G, H, W, B, layers, skip=True):
    from lic360_fused import FusedCodec
    old = os.environ.pop("LIC360_NOSKIP", None)
    if not skip:
        os.environ["LIC360_NOSKIP"] = "1"
    try:
        fc = FusedCodec(G, H, W, max_batch=B)
    finally:
        os.environ.pop("LIC360_NOSKIP", None)
        if old is not None:
            os.environ["LIC360_NOSKIP"] = old
    fc.load_layers(layers)
    return fc


def _batch(G, H, W, B, seed, n_iid=2):
    cs, ms = [], []
    for i in range(B):
        c, m, _ = (latent if i >= B - n_iid else latent_smooth)(np.random.default_rng(seed + i), G, H, W)
        cs.append(c)
        ms.append(m)
    return np.concatenate(cs, 0), np.concatenate(ms, 0)


def test_full_size_smooth_goldens_with_poisoned_buffers():
    """cfg2s / cfg3s (oracle bytes of SURVEY 8d's smooth masks) as images 0 of a batch of 16: encode == oracle bytes, decode of the oracle's bytes ==
    the symbols -- with the skip on in both orders (16 images: list-mode decode) and 1e10 in every activation cell beforehand"""
    g2, g3 = np.load(os.path.join(GOLD, "full_cfg2s.npz")), np.load(os.path.join(GOLD, "full_cfg3s.npz"))
    for g in (g2, g3):
        G, H, W = int(g["G"]), int(g["H"]), int(g["W"])
        code, mask, _ = latent_smooth(np.random.default_rng(int(g["latent_seed"])), G, H, W)
        assert hashlib.sha256(code.tobytes()).hexdigest() == str(g["code_sha256"]) and hashlib.sha256(mask.tobytes()).hexdigest() == str(g["mask_sha256"])
        layers = make_main_params(int(g["weight_seed"]), G)
        c15, m15 = _batch(G, H, W, 15, 777)
        code16, mask16 = np.concatenate([code, c15], 0), np.concatenate([mask, m15], 0)
        fc = _codec(G, H, W, 16, layers)
        assert fc.skip_active() == 2
        fc.debug_fill(POISON)
        streams = fc.encode(dev(code16), dev(mask16))
        assert streams[0] == g["bytes"].tobytes()
        fc.debug_fill(-POISON)
        out = fc.decode([g["bytes"].tobytes()] + streams[1:], dev(mask16)).cpu().numpy()
        assert np.array_equal(out, code16 * mask16)


@pytest.mark.parametrize("G,H,W,B", [(48, 16, 24, 16), (12, 64, 20, 24), (48, 64, 128, 16)])
def test_skip_changes_no_byte(G, H, W, B):
    code, mask = _batch(G, H, W, B, 4242 + H)
    layers = make_main_params(99 + G, G)
    ref = _codec(G, H, W, B, layers, skip=False)
    assert ref.skip_active() == 0
    want = ref.encode(dev(code), dev(mask))
    fc = _codec(G, H, W, B, layers)
    assert fc.skip_active() == 2
    fc.debug_fill(POISON)
    fc.skip_stats(True)
    got = fc.encode(dev(code), dev(mask))
    assert got == want
    fc.debug_fill(POISON)
    out = fc.decode(want, dev(mask)).cpu().numpy()
    assert np.array_equal(out, code * mask)
    assert np.array_equal(ref.decode(want, dev(mask)).cpu().numpy(), code * mask)
    # a second pass over DIFFERENT masks in the same buffers: what is dead now was live (and holds values) before
    code2, mask2 = _batch(G, H, W, B, 999 + H, n_iid=B // 2)
    got2 = fc.encode(dev(code2), dev(mask2))
    assert got2 == ref.encode(dev(code2), dev(mask2))
    assert np.array_equal(fc.decode(got2, dev(mask2)).cpu().numpy(), code2 * mask2)
    enc, dec = fc.skip_stats(False, read=True)
    assert enc.sum() > 0 and dec.sum() > 0


def _need_of(mask):
    return np.stack([brute_need(m > 0.5) for m in mask], 0) if mask.shape[-1] * mask.shape[-2] <= 512 else None


@pytest.mark.parametrize("G,H,W,B", [(12, 16, 24, 16), (48, 12, 16, 32)])
def test_task_lists_cover_the_live_cells_exactly_once(G, H, W, B):
    code, mask = _batch(G, H, W, B, 31337)
    layers = make_main_params(5, G)
    fc = _codec(G, H, W, B, layers)
    streams = fc.encode(dev(code), dev(mask))
    fc.decode(streams, dev(mask))
    need, _ = fc.debug_lists(0)
    need = need.reshape(B, 12, H, W).astype(np.int64)
    assert np.array_equal(need, _need_of(mask))
    S, P = H + W - 1, H + W + G - 2
    # ---- decode order
    cnt, cap = fc.debug_lists(3)
    rec, _ = fc.debug_lists(4)
    cnt, rec = cnt.reshape(12, P, 8), rec.reshape(12, P, 8, cap, 4)
    stored = np.zeros((12, 3 * B, G, H, W), np.int32)
    for l in range(1, 12):
        for p in range(P):
            for x in range(8):
                for r in rec[l, p, x, :cnt[l, p, x]]:
                    g0, packed, n = int(r[0] & 127), int((r[0] >> 7) & 7), int(r[0] >> 10)
                    gm = int(r[1] >> 22) & 7                                    # the groups of the block that the record computes at all
                    assert gm
                    assert n < 3 * B and g0 % 3 == 0                            # (n % 8 is the record's HOME XCD; the balancing pass may have moved it to list x)
                    pieces = []
                    if packed == 0:
                        pieces.append((n, 0, H - 1, None))
                    else:
                        assert packed == 4
                        last = None
                        for w in r[1:]:
                            w = int(w)
                            if not w >> 21:
                                continue
                            k, slo, shi, a0 = w & 7, (w >> 3) & 63, (w >> 9) & 63, (w >> 15) & 63
                            assert slo <= shi < H and (a0 - slo) % 4 == 0 and a0 + shi - slo <= 63
                            assert a0 >= 2 or slo == 0
                            assert a0 + shi - slo <= 61 or shi == H - 1
                            if last is not None:
                                assert a0 >= ((last + 4) // 4 + 1) * 4          # the neighbour's band columns (lane .. lane + 4) lie in other quads
                            last = a0 + shi - slo
                            assert (n + 8 * k) // B == n // B                   # one net per record: its waves' weights are uniform
                            pieces.append((n + 8 * k, slo, shi, a0))
                        assert pieces
                    for smp, slo, shi, _ in pieces:
                        for q in range(3):
                            g, s = g0 + q, p - g0 - q
                            if g >= G or s < 0 or s >= S or not gm >> q & 1:
                                continue
                            ys = np.arange(max(slo, s - W + 1, 0), min(shi, s, H - 1) + 1)
                            stored[l, smp, g, ys, s - ys] += 1
    assert stored.max() <= 1
    live = np.arange(G)[None, None, :, None, None] <= need.transpose(1, 0, 2, 3)[:, :, None]      # [12, B, G, H, W]
    for net in range(3):
        assert not (live[1:] & (stored[1:, net * B:(net + 1) * B] == 0)).any()
    # ---- encode order: hidden layers (4 groups x 4 tiles) and the fused last layer (5 groups x 2 tiles)
    ecnt, ecap = fc.debug_lists(1)
    elist, _ = fc.debug_lists(2)
    ecnt, elist = ecnt.reshape(12, 8), elist.reshape(12, 8, ecap)
    ntx, nty = (W + 15) // 16, (H + 3) // 4
    ntiles = ntx * nty
    tmax = np.full((B, 12, nty * 4, ntx * 16), -1, np.int64)
    tmax[:, :, :H, :W] = need
    tmax = tmax.reshape(B, 12, nty, 4, ntx, 16).max((3, 5)).reshape(B, 12, ntiles)
    for l in range(1, 12):
        gpb, tpt, N = (5, 2, B) if l == 11 else (4, 4, 3 * B)
        n_gb, n_chunks, gbk = (G + gpb - 1) // gpb, (ntiles + tpt - 1) // tpt, 16
        seen = np.zeros((N, ntiles, n_gb), np.int32)
        for x in range(8):
            ns_x = (N - x + 7) >> 3
            units = ns_x * n_chunks
            us = [int(e & 0x0fffffff) for e in elist[l, x, :ecnt[l, x]].view(np.uint32)]
            assert us == sorted(us) and len(set(us)) == len(us)               # launch order kept
            for e in elist[l, x, :ecnt[l, x]].view(np.uint32):
                u, tm = int(e & 0x0fffffff), int(e >> 28)
                per = gbk * n_gb
                blk, r = divmod(u, per)
                kk = min(units - blk * gbk, gbk)
                gb = n_gb - 1 - r // kk
                v = blk * gbk + r % kk
                tile0, n = (v % n_chunks) * tpt, x + 8 * (v // n_chunks)
                assert tm and n < N
                for t in range(tpt):
                    if tm >> t & 1:
                        assert tile0 + t < ntiles
                        seen[n, tile0 + t, gb] += 1
        want = tmax[np.arange(N) % B, l][:, :, None] >= (np.arange(n_gb) * gpb)[None, None, :]
        assert np.array_equal(seen, want.astype(np.int32))
