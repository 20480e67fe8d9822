"""Achieved HBM GB/s of the streaming ops of the path (algorithmic bytes / time against 8 TB/s), at the
512x1024 single-image sizes of test/model_zoo.py and at a 32-image batch.  Writes one JSON document to stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "360-image-compression_amd"))
import torch  # noqa: E402
import lic360  # noqa: E402

PEAK = 8000.0


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def selfcheck(device=0):
    """every timed op once against an independent torch statement of the same map, at the timed shapes (no oracle here: the oracle
    comparison at these shapes lives in tests/test_gpu_ops.py::test_*_at_bench_shapes) -- a wrong kernel must not be timed"""
    dev = "cuda:%d" % device
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((2, 192, 256, 512), device=dev, generator=g)
    mid = torch.cat([x[..., -2:], x, x[..., :2]], -1)
    want = torch.cat([mid[:, :, :2].flip(2, 3), mid, mid[:, :, -2:].flip(2, 3)], 2)           # lon wrap, pole rows reflected + mirrored
    y = want.clone()
    y[:, :, :2] = 7; y[:, :, -2:] = 7; y[..., :2] = 7; y[..., -2:] = 7
    lic360.SpherePadOp(2, True, device, False).forward(y)
    assert torch.equal(y, want), "sphere_pad in place"
    lic360.SphereTrimOp(2, device, False).forward(y)
    z = want.clone()
    z[:, :, :2] = 0; z[:, :, -2:] = 0; z[..., :2] = 0; z[..., -2:] = 0
    assert torch.equal(y, z), "sphere_trim"
    v = torch.randn((2, 192, 32, 64), device=dev, generator=g)
    mid = torch.cat([v[..., -2:], v, v[..., :2]], -1)
    assert torch.equal(lic360.SpherePadOp(2, False, device, False).forward(v)[0], torch.cat([mid[:, :, :2].flip(2, 3), mid, mid[:, :, -2:].flip(2, 3)], 2)), "sphere_pad"
    assert torch.equal(lic360.DtowOp(2, True, device, False).forward(v)[0], torch.nn.functional.pixel_shuffle(v, 2)), "dtow"
    wb = torch.randn((192, 8), device=dev, generator=g).sort(dim=1).values.contiguous()
    centres = torch.cat([wb[:, :1], wb[:, :1] + torch.cumsum(torch.exp(wb[:, 1:]), 1)], 1)                # quant_cuda.cu:35-43
    top, qi = lic360.QuantOp(192, 8, 0.9, 100, 2, 0.1, device, False).forward(v, wb, torch.zeros((192, 8), device=dev), False)
    picked = torch.gather(centres[None, :, None, None, :].expand(2, 192, 32, 64, 8), 4, qi.long()[..., None])[..., 0]
    d = (v[..., None] - centres[None, :, None, None, :]).abs()
    assert torch.allclose(top, picked, rtol=1e-4, atol=1e-5), "quant: value = centre of the returned index"
    assert float(((v - top).abs() > d.min(-1).values * (1 + 1e-4) + 1e-5).float().mean()) < 1e-3, "quant picks the nearest centre"
    msk = (torch.rand(v.shape, device=dev, generator=g) > 0.3).float()
    dq = lic360.DquantOp(192, 8, device, False).forward(qi, msk, wb)[0]
    assert torch.allclose(dq, torch.where(msk > 0.5, picked, centres[None, :, None, None, 0].expand_as(picked)), rtol=1e-4, atol=1e-5), "dquant"
    return True


def tile_rows(n, device=0, G=48, H=64, W=128):
    """the plane gather / scatter / add operators of the per-plane drivers (A11-A13: tile_extract[_batch], tile_input, tile_add) over the WHOLE
    238-plane sweep of a 48 x 64 x 128 latent, n images per call, through the C ABI: algorithmic bytes = 8 B per element moved (tile_add: 12 B,
    two reads and a write) summed over the planes' real lengths, time = the whole sweep of 238 launches (they are launch-bound: <= 3072 positions
    per image and plane).  Checked against torch index arithmetic at these shapes before timing."""
    import ctypes as C
    L = lic360._lib
    dev = "cuda:%d" % device
    st = lambda: lic360._stream(device)
    HW, P = H * W, H + W + G - 2
    idx_h, pidx_h = (C.c_int * (2 * HW))(), (C.c_int * (H + W))()
    lic360._chk(L.lic360_code_contex(H, W, idx_h, pidx_h))
    idx = torch.tensor(list(idx_h), dtype=torch.int32, device=dev)
    win = []
    for p in range(P):
        a, b = C.c_int(0), C.c_int(0)
        lic360._chk(L.lic360_plane_window(p, G, H, W, pidx_h, C.byref(a), C.byref(b)))
        win.append((a.value, b.value))
    tot = sum(ln for _, ln in win)
    assert tot == G * HW
    g = torch.Generator(device=dev).manual_seed(3)
    lab = torch.randn((n, G, H, W), device=dev, generator=g)                      # TileExtract(label / mask): cpn = 1
    y3 = torch.randn((3 * n, 3 * G, H, W), device=dev, generator=g)               # TileExtractBatch: the three nets' 144 outputs, cpn = 3
    act = torch.randn((3 * n, 4 * G, H, W), device=dev, generator=g)              # TileAdd: 192-channel activations of the three nets
    act2 = torch.randn((3 * n, 4 * G, H, W), device=dev, generator=g)
    sym = torch.randn((n, 1, H, W), device=dev, generator=g)                      # TileInput: compact symbols of the previous plane
    out1 = torch.zeros((n, 1, H, W), device=dev)
    out3 = torch.zeros((3 * n, 3, H, W), device=dev)
    outi = torch.zeros((3 * n, G, H, W), device=dev)
    th, tw = idx[:HW].long(), idx[HW:].long()
    # ---- checks on one mid-sweep plane (tile_extract_cuda.cu:36-41, :107-114; tile_input_cuda.cu:33-40; tile_add_cuda.cu:28-34)
    p = 100
    s0, ln = win[p]
    hh, ww = th[s0:s0 + ln], tw[s0:s0 + ln]
    gg = p - hh - ww
    lic360._chk(L.lic360_tile_extract(st(), lic360._p(lab), lic360._p(out1), n, G, H, W, G, lic360._p(idx), s0, ln, p))
    assert torch.equal(out1.flatten()[:n * ln].view(n, ln), lab[:, gg, hh, ww]), "tile_extract"                   # packed [n][len][cpn = 1]
    lic360._chk(L.lic360_tile_extract_batch(st(), lic360._p(y3), lic360._p(out3), 3 * n, 3 * G, H, W, G, lic360._p(idx), s0, ln, p))
    want = torch.stack([y3[:, gg * 3 + c, hh, ww] for c in range(3)], -1)          # [3n, ln, 3]
    got = out3.flatten().view(3, -1)[:, :n * ln * 3].view(3, n, ln, 3)              # three slabs (stride cpn h w n), each packed [n][len][cpn = 3]
    assert torch.equal(got, want.view(3, n, ln, 3)), "tile_extract_batch"
    lic360._chk(L.lic360_tile_input(st(), lic360._p(sym), lic360._p(outi), n, G, H, W, lic360._f(-3.5), lic360._f(1.0), 3, lic360._p(idx), s0, ln, p))
    assert torch.equal(outi[:, gg, hh, ww], (sym.flatten()[:n * ln].view(n, ln) * 1.0 + -3.5).repeat(3, 1)), "tile_input"   # symbols packed [n][len]
    ref = act.clone()
    for c in range(4):
        ref[:, gg * 4 + c, hh, ww] += act2[:, gg * 4 + c, hh, ww]
    a2 = act.clone()
    lic360._chk(L.lic360_tile_add(st(), lic360._p(a2), lic360._p(act2), 3 * n, 4 * G, H, W, G, lic360._p(idx), s0, ln, p))
    assert torch.equal(a2, ref), "tile_add"
    del a2, ref

    def sweep(fn):
        def run():
            for p, (s0, ln) in enumerate(win):
                if ln > 0:
                    fn(p, s0, ln)
        return run
    ops = [
        ("tile_extract (label / mask plane gather, 238-plane sweep)", 8.0 * n * tot,
         sweep(lambda p, s0, ln: L.lic360_tile_extract(st(), lic360._p(lab), lic360._p(out1), n, G, H, W, G, lic360._p(idx), s0, ln, p))),
        ("tile_extract_batch (3 nets x 3 GMM parameters, 238-plane sweep)", 8.0 * 9 * n * tot,
         sweep(lambda p, s0, ln: L.lic360_tile_extract_batch(st(), lic360._p(y3), lic360._p(out3), 3 * n, 3 * G, H, W, G, lic360._p(idx), s0, ln, p))),
        ("tile_input (symbol scatter x3 nets, 238-plane sweep)", 4.0 * 4 * n * tot,
         sweep(lambda p, s0, ln: L.lic360_tile_input(st(), lic360._p(sym), lic360._p(outi), n, G, H, W, lic360._f(-3.5), lic360._f(1.0), 3, lic360._p(idx), s0, ln, p))),
        ("tile_add (residual add on the plane, 3 nets x 4 channels, 238-plane sweep)", 12.0 * 12 * n * tot,
         sweep(lambda p, s0, ln: L.lic360_tile_add(st(), lic360._p(act), lic360._p(act2), 3 * n, 4 * G, H, W, G, lic360._p(idx), s0, ln, p))),
    ]
    rows = []
    for name, nbytes, fn in ops:
        t = timed(fn, reps=5)
        rows.append({"op": name, "images": n, "algorithmic_MB": nbytes / 1e6, "us": t * 1e6, "GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / 1e9 / PEAK,
                     "launches": len([1 for _, ln in win if ln > 0]), "us_per_launch": t * 1e6 / P})
    return rows


def measure(batches=(1, 32), device=0):
    """-> rows of {op, images, algorithmic_MB, us, GBps, frac_of_hbm_peak}; called by bench.py (batch 32) and by __main__"""
    dev = "cuda:%d" % device
    rows = []
    selfcheck(device)

    def add(name, n, nbytes, fn):
        t = timed(fn)
        rows.append({"op": name, "images": n, "algorithmic_MB": nbytes / 1e6, "us": t * 1e6, "GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / 1e9 / PEAK})

    for n in batches:
        g = torch.Generator(device=dev).manual_seed(1)
        # ERP apron refresh on the largest feature map (in place: only the 2-cell apron is read and written)
        x = torch.randn((n, 192, 260, 516), device=dev, generator=g)
        apron = n * 192 * (260 * 516 - 256 * 512) * 4
        pad = lic360.SpherePadOp(2, True, device, False)
        add("sphere_pad (in place, 260x516)", n, 2 * apron, lambda: pad.forward(x))
        trim = lic360.SphereTrimOp(2, device, False)
        add("sphere_trim (in place, 260x516)", n, apron, lambda: trim.forward(x))
        del x
        x = torch.randn((n, 192, 260, 516), device=dev, generator=g)
        tp = lambda: lic360._chk(lic360._lib.lic360_sphere_trim_pad_inplace(lic360._stream(device), lic360._p(x), n * 192, 260, 516, 2))
        add("sphere_trim + sphere_pad in place as one pass (260x516)", n, 2 * apron, tp)
        del x
        # out-of-place pad of the decoder's first map
        y = torch.randn((n, 192, 32, 64), device=dev, generator=g)
        pad2 = lic360.SpherePadOp(2, False, device, False)
        add("sphere_pad (36x68 from 32x64)", n, n * 192 * (32 * 64 + 36 * 68) * 4, lambda: pad2.forward(y))
        # pixel (un)shuffle, quantiser, dequantiser on the latent
        dtow = lic360.DtowOp(2, True, device, False)
        add("dtow [192,32,64]->[48,64,128]", n, 2 * y.numel() * 4, lambda: dtow.forward(y))
        wb = torch.randn((192, 8), device=dev, generator=g).sort(dim=1).values.contiguous()
        cnt = torch.zeros((192, 8), device=dev)
        q = lic360.QuantOp(192, 8, 0.9, 100, 2, 0.1, device, False)
        add("quant (value + index out)", n, 3 * y.numel() * 4, lambda: q.forward(y, wb, cnt, False))
        idx = torch.randint(0, 8, y.shape, device=dev, generator=g).float()
        msk = (torch.rand(y.shape, device=dev, generator=g) > 0.3).float()
        dq = lic360.DquantOp(192, 8, device, False)
        add("dquant", n, 3 * y.numel() * 4, lambda: dq.forward(idx, msk, wb))
        # drop-in GMM table op: M symbols x (9 floats in, 9 floats out)
        m = n * 8192
        w3, d3, m3 = (torch.randn((m, 3, 1, 1), device=dev, generator=g) for _ in range(3))
        tab = lic360.EntropyGmmTableOp(8, 3.5, 3, 65536, 1e-6, device, False)
        tn = torch.tensor([m], dtype=torch.int32)
        add("entropy_gmm_table (8192 symbols per image)", n, m * 72, lambda: tab.forward(w3, d3, m3, tn))      # (softmax / sigma floor are written back in place, as in the reference)
        rows.extend(tile_rows(n, device))
        # the rows SURVEY.md 8f.3 / 8f.4 widened into: viewport projection of decoded RGB images, two training-side gradients
        img = torch.rand((n, 3, 512, 1024), device=dev, generator=g)
        pr = lic360.ProjectsOp(171, 256, [-0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, 0, 0],
                               [0, 0, 0, 0, 0.25, 0.25, 0.25, 0.25, -0.25, -0.25, -0.25, -0.25, 0.5, -0.5], 0.5, False, device, False)
        vp = n * 14 * 3 * 171 * 256 * 4
        add("projects forward (14 viewports 171x256 of 512x1024 RGB)", n, vp + n * 14 * 171 * 256 * 8 + img.numel() * 4, lambda: pr.forward(img))
        gp = torch.randn((n, 192, 36, 68), device=dev, generator=g)
        add("sphere_pad backward (36x68 -> 32x64)", n, n * 192 * (32 * 64 + 36 * 68) * 4, lambda: pad2.backward(gp))
        q.forward(y, wb, cnt, True)
        gq = torch.randn(y.shape, device=dev, generator=g)
        add("quant backward (data + level gradients)", n, 5 * y.numel() * 4, lambda: q.backward([gq, gq], y, q._top[0]))
    return rows


if __name__ == "__main__":
    rows = measure(batches=tuple(int(b) for b in os.environ.get("SOB_BATCHES", "1,32").split(",")))
    print(json.dumps({"peak_GBps": PEAK, "note": "torch.cuda.Event timing of 50 back-to-back calls through the lic360 shim (includes its per-call Python/ctypes "
                  "overhead; single-image tensors are a few MB, so those rows are launch-latency bound)", "rows": rows}, indent=1))
