"""debug helper: run lic360_cconv16_ec cases against the oracle and summarise where they differ"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import lic360 as lic, oracle as orc
from util import conv_params
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
cases = [(6, 1, 4, False, True, 3, 3, 7, 70), (7, 1, 4, False, False, 1, 1, 33, 40)]
import itertools
for case, (use_res, use_zero, rep) in itertools.product(cases, [(0,0,0),(1,0,0),(0,1,0),(1,1,0),(1,1,1)]):
    G, cin, cout, hidden, act, nb, N, H, W = case
    import zlib; rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    Cc, nout = G * cin, G * cout
    w, b, a = conv_params(rng, nb if nb > 1 else None, nout, Cc, act=act)
    if nb == 1:
        w, b = w[None], b[None]
        a = None if a is None else a[None]
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    if use_zero: x[rng.random(x.shape) < 0.2] = 0.0
    res = rng.standard_normal((N, nout, H, W)).astype(np.float32)
    constrain = 6 if hidden else 5
    ref = orc.cconv_ec(x, w, b, a, G, constrain) + (res if use_res else 0)
    L = lic._lib
    hp, wp = C.c_int(), C.c_int()
    L.lic360_ec16_layout(H, W, C.byref(hp), C.byref(wp)); hp, wp = hp.value, wp.value
    pad = lambda t: np.pad(t, ((0, 0), (0, 0), (2, hp - H - 2), (2, wp - W - 2)))
    plan = C.c_void_p(0)
    L.lic360_conv_plan_create(Cc, G, nout, 5, constrain, C.byref(plan))
    packed = torch.empty(nb * L.lic360_conv16_packed_floats(plan), dtype=torch.float32, device="cuda:0")
    xd, wd, bd, rd = dev(pad(x)), dev(w), dev(b), dev(pad(res))
    ad = dev(a) if act else None
    out = torch.zeros((N, nout, hp, wp), dtype=torch.float32, device="cuda:0")
    ctr = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    s, P = lic._stream(0), lic._p
    assert L.lic360_conv16_pack(s, plan, P(wd), nb, P(packed)) == 0
    assert L.lic360_cconv16_ec(s, plan, P(xd), P(packed), P(bd), P(ad), P(rd) if use_res else None, P(out), N, H, W, nb, N, P(ctr)) == 0
    got = out.cpu().numpy()[:, :, 2:2 + H, 2:2 + W]
    bad = got != ref
    print(case, use_res, use_zero, rep, "mismatch frac %.4f" % bad.mean(), "max abs", np.abs(got - ref).max())
    if bad.any():
        print("  per sample:", bad.mean(axis=(1, 2, 3)))
        print("  per group :", np.round(bad.reshape(N, G, cout, H, W).mean(axis=(0, 2, 3, 4)), 2))
        print("  per r     :", np.round(bad.reshape(N, G, cout, H, W).mean(axis=(0, 1, 3, 4)), 2))
        print("  per row   :", np.round(bad.mean(axis=(0, 1, 3)), 2))
        print("  per col16 :", np.round(bad.mean(axis=(0, 1, 2))[:min(W, 40)], 1))
        close = np.abs(got - ref) < 1e-4
        print("  close frac (1e-4):", close.mean())

