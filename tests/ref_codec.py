"""Oracle-side restatement of the codec drivers D1-D3 (test/lic360_demo.py:95-290) composed from the
CPU oracle's ops; used by the GPU parity tests, smoke() and bench.py's cpu_baseline leg."""
import numpy as np

import oracle as orc
from util import conv_params, make_main_params, make_imp_params  # noqa: F401  (weight synthesis lives in util: no oracle needed)


def load_into_driver(drv, layers):
    """Copy oracle-format parameters into an EntEncoderFast / EntDecoder / Imp* driver (net.N.* keys)."""
    import torch
    mods = [drv.net[0]]
    for i in range(1, 6):
        mods += [drv.net[i].conv1, drv.net[i].conv2]
    mods.append(drv.net[6])
    dev = drv.cuda_name
    for m, l in zip(mods, layers):
        m.weight.data = torch.from_numpy(l["w"]).to(dev)
        m.bias.data = torch.from_numpy(l["b"]).to(dev)
        if l["a"] is not None:
            m.relu.data = torch.from_numpy(l["a"]).to(dev)


def net_ec(x, layers, G):
    y = orc.cconv_ec(x, layers[0]["w"], layers[0]["b"], layers[0]["a"], G, layers[0]["constrain"])
    for i in range(5):
        l1, l2 = layers[1 + 2 * i], layers[2 + 2 * i]
        t = orc.cconv_ec(y, l1["w"], l1["b"], l1["a"], G, 6)
        t = orc.cconv_ec(t, l2["w"], l2["b"], l2["a"], G, 6)
        y = t + y
    l = layers[11]
    return orc.cconv_ec(y, l["w"], l["b"], l["a"], G, 6)


def encode_main(code, mask, layers, G, want_tables=False):
    """D1 EntEncoderFast.forward -> bitstream bytes."""
    _, _, H, W = code.shape
    idx, pidx = orc.code_contex(H, W)
    t = ((code - np.float32(3.5)) * mask).astype(np.float32)
    y = net_ec(np.concatenate([t, t, t], 0), layers, G)
    enc = orc.Encoder()
    z = np.zeros(3 * 3 * H * W, np.float32)
    lab = np.zeros(H * W, np.float32)
    mk = np.zeros(H * W, np.float32)
    tabs = []
    for p in range(H + W + G - 2):
        tn = orc.tile_extract_batch(y, z, G, idx, pidx, p)
        tab = orc.gmm_table_batch(z, 3 * H * W, tn)
        orc.tile_extract(code, lab, G, True, idx, pidx, p)
        orc.tile_extract(mask, mk, G, True, idx, pidx, p)
        enc.encode(tab.astype(np.int32), 8, lab[:tn].astype(np.int32), mk[:tn], tn)
        if want_tables:
            tabs.append((tab.astype(np.int32), lab[:tn].astype(np.int32).copy(), mk[:tn].copy()))
    data = enc.finish()
    return (data, tabs) if want_tables else data


class _PlaneNet:
    """DC net state: persistent per-layer outputs (op-owned buffers in the reference)."""

    def __init__(self, layers, G, N, H, W):
        self.layers, self.G = layers, G
        self.out = [np.zeros((N, l["w"].shape[-4], H, W), np.float32) for l in layers]

    def step(self, x, idx, pidx, p):
        G, L, o = self.G, self.layers, self.out
        orc.cconv_dc_plane(x, L[0]["w"], L[0]["b"], L[0]["a"], o[0], G, L[0]["constrain"], idx, pidx, p)
        y = o[0]
        for i in range(5):
            a, b = 1 + 2 * i, 2 + 2 * i
            orc.cconv_dc_plane(y, L[a]["w"], L[a]["b"], L[a]["a"], o[a], G, 6, idx, pidx, p)
            orc.cconv_dc_plane(o[a], L[b]["w"], L[b]["b"], L[b]["a"], o[b], G, 6, idx, pidx, p)
            orc.tile_add(o[b], y, G, idx, pidx, p)
            y = o[b]
        orc.cconv_dc_plane(y, L[11]["w"], L[11]["b"], L[11]["a"], o[11], G, 6, idx, pidx, p)
        return o[11]


def decode_main(data, mask, layers, G):
    """D2 EntDecoder.forward -> code tensor [1,G,H,W]."""
    _, _, H, W = mask.shape
    idx, pidx = orc.code_contex(H, W)
    dec = orc.Decoder(data)
    net = _PlaneNet(layers, G, 3, H, W)
    b = np.zeros((3, G, H, W), np.float32)
    pout = np.zeros(H * W, np.float32)
    z = np.zeros(3 * 3 * H * W, np.float32)
    mk = np.zeros(H * W, np.float32)
    for p in range(H + W + G - 2):
        orc.tile_input(pout, b.reshape(-1), 1, G, H, W, -3.5, 1.0, 3, idx, pidx, p)
        y = net.step(b, idx, pidx, p)
        tn = orc.tile_extract_batch(y, z, G, idx, pidx, p)
        tab = orc.gmm_table_batch(z, 3 * H * W, tn)
        orc.tile_extract(mask, mk, G, True, idx, pidx, p)
        pout = dec.decode(tab.astype(np.int32), 8, mk[:tn], tn, 3.5, size=H * W)
    orc.tile_input(pout, b.reshape(-1), 1, G, H, W, -3.5, 1.0, 3, idx, pidx, H + W + G - 2)
    dec.close()
    return (b[0:1] + np.float32(3.5) * mask).astype(np.float32)


def encode_imp(levels, layers, nsym=49):
    """D3 ImpEntEncoderFast.forward: levels [1,1,h,w] in {0..48} -> bytes."""
    _, _, H, W = levels.shape
    idx, pidx = orc.code_contex(H, W)
    x = orc.scale(levels, -1.0, np.float32(2.0 / (nsym - 2)))
    y = net_ec(x, layers, 1)
    enc = orc.Encoder()
    z = np.zeros(nsym * H * W, np.float32)
    lab = np.zeros(H * W, np.float32)
    for p in range(H + W - 1):
        tn = orc.tile_extract(y, z, 1, True, idx, pidx, p)
        tab = orc.entropy_table(z, tn, nsym)
        orc.tile_extract(levels, lab, 1, True, idx, pidx, p)
        enc.encode(tab.astype(np.int32), nsym, lab[:tn].astype(np.int32), None, tn)
    return enc.finish()


def decode_imp(data, layers, H, W, nsym=49):
    """D3 ImpEntDecoder.forward -> levels [1,1,H,W] (before Imp2mask/Dtow)."""
    idx, pidx = orc.code_contex(H, W)
    dec = orc.Decoder(data)
    net = _PlaneNet(layers, 1, 1, H, W)
    sc = np.float32(2.0 / (nsym - 2))
    b = np.zeros((1, 1, H, W), np.float32)
    pout = np.zeros(H * W, np.float32)
    z = np.zeros(nsym * H * W, np.float32)
    for p in range(H + W - 1):
        orc.tile_input(pout, b.reshape(-1), 1, 1, H, W, -1.0, sc, 1, idx, pidx, p)
        y = net.step(b, idx, pidx, p)
        tn = orc.tile_extract(y, z, 1, True, idx, pidx, p)
        tab = orc.entropy_table(z, tn, nsym)
        pout = dec.decode(tab.astype(np.int32), nsym, None, tn, 3.5, size=H * W)
    orc.tile_input(pout, b.reshape(-1), 1, 1, H, W, -1.0, sc, 1, idx, pidx, H + W - 1)
    dec.close()
    code = ((b + np.float32(1.0)) / sc).astype(np.float32)
    return np.floor(code + np.float32(1e-5)).astype(np.float32)
