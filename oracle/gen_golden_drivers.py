#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY: run the reference's OWN codec drivers (test/lic360_demo.py:95-322 -- EntEncoderFast, EntDecoder,
ImpEntEncoderFast, ImpEntDecoder, cast_entropy_parameter, cast_imp_entropy_parameter) on the CPU and write what they produce to
tests/golden/driver_*.npz (inputs, seeds, bitstreams, decoded tensors: data only).

The reference's Python is imported from /root/reference where it lies, never copied or edited:
  * its compiled extension `lic360` is replaced by oracle/ref_backend/lic360.py (the extension's per-op state over the CPU oracle's
    kernels; the reference's ArithmeticCoder.cpp re-encodes every bitstream as a cross-check);
  * the drivers hard-code 'cuda:N' (lic360_demo.py:101,150,197,248 and the wrappers' set_param): this script maps every cuda device
    to the CPU by wrapping torch.Tensor.to / torch.nn.Module.to before the import, and gives each wrapper's op dictionary the key
    `x.device.index` has on the CPU (None) beside its device id;
  * `cv2` / `tkinter` (absent here, imported at module scope by lic360_demo.py:7 / lic360_operator/Dquant.py:1) get bare placeholders.
Weights: the seeded synthetic parameters of tests/util.py, laid out as a TRAINING checkpoint (`ent.weight_net.* / ent.delta_net.* /
ent.mean_net.*`, `imp_ent.net.*`) and pushed through the reference's cast_* functions -- so the key mapping and the
[weight, sigma, mu] batch order are the reference's, not ours.

tests/test_driver_golden.py (CPU: tests/ref_codec.py; -m gpu: the op-level drivers and the fused codecs) must reproduce the files.
usage: python3 oracle/gen_golden_drivers.py [a b c]    (needs /root/reference, oracle/liblic360_oracle.so; cases a, b ~1 min, c ~10 min)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def map_cuda_to_cpu():
    def m(a):
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        return a
    t_to, m_to = torch.Tensor.to, torch.nn.Module.to
    torch.Tensor.to = lambda self, *a, **k: t_to(self, *[m(x) for x in a], **{kk: m(v) for kk, v in k.items()})
    torch.nn.Module.to = lambda self, *a, **k: m_to(self, *[m(x) for x in a], **{kk: m(v) for kk, v in k.items()})


def import_reference_demo():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    if "tkinter" not in sys.modules:
        try:
            import tkinter.messagebox  # noqa: F401
        except Exception:                                          # noqa: BLE001
            tk, mb = types.ModuleType("tkinter"), types.ModuleType("tkinter.messagebox")
            mb.NO = "no"
            tk.messagebox = mb
            sys.modules["tkinter"], sys.modules["tkinter.messagebox"] = tk, mb
    sys.path[:0] = [os.path.join(ROOT, "oracle", "ref_backend"), os.path.join(ROOT, "oracle"), REF, os.path.join(REF, "test"),
                    os.path.join(ROOT, "tests")]
    import lic360
    assert os.path.dirname(os.path.abspath(lic360.__file__)) == os.path.join(ROOT, "oracle", "ref_backend")
    import lic360_operator
    assert os.path.dirname(os.path.abspath(lic360_operator.__file__)) == os.path.join(REF, "lic360_operator")
    import lic360_demo
    assert os.path.abspath(lic360_demo.__file__) == os.path.join(REF, "test", "lic360_demo.py")
    return lic360_demo


def cpu_keys(module):
    """x.device.index is None on the CPU: every wrapper's {device id: op} dictionary answers to both keys"""
    for mod in module.modules():
        op = getattr(mod, "op", None)
        if isinstance(op, dict) and op:
            one = next(iter(op.values()))
            op[None] = one
            for gid in getattr(mod, "device_list", [0]):
                op[gid] = one
    return module


def checkpoint_main(layers):
    """training-side key layout of the three GMM-parameter nets (what cast_entropy_parameter reads, lic360_demo.py:296-310)"""
    ck = {}
    for b, pre in enumerate(("ent.weight_net", "ent.delta_net", "ent.mean_net")):
        def put(layer, wk, bk, rk):
            ck[wk] = torch.from_numpy(layers[layer]["w"][b].copy())
            ck[bk] = torch.from_numpy(layers[layer]["b"][b].copy())
            if rk is not None:
                ck[rk] = torch.from_numpy(layers[layer]["a"][b].copy())
        put(0, pre + ".0.weight", pre + ".0.bias", pre + ".1.weight")
        for bid in range(1, 6):
            put(2 * bid - 1, "%s.%d.net.0.weight" % (pre, bid + 1), "%s.%d.net.0.bias" % (pre, bid + 1), "%s.%d.net.1.weight" % (pre, bid + 1))
            put(2 * bid, "%s.%d.net.2.weight" % (pre, bid + 1), "%s.%d.net.2.bias" % (pre, bid + 1), "%s.%d.net.3.weight" % (pre, bid + 1))
        put(11, pre + ".7.weight", pre + ".7.bias", None)
    return ck


def checkpoint_imp(layers):
    ck, pre = {}, "imp_ent.net"

    def put(layer, wk, bk, rk):
        ck[wk] = torch.from_numpy(layers[layer]["w"].copy())
        ck[bk] = torch.from_numpy(layers[layer]["b"].copy())
        if rk is not None:
            ck[rk] = torch.from_numpy(layers[layer]["a"].copy())
    put(0, pre + ".0.weight", pre + ".0.bias", pre + ".1.weight")
    for bid in range(1, 6):
        put(2 * bid - 1, "%s.%d.net.0.weight" % (pre, bid + 1), "%s.%d.net.0.bias" % (pre, bid + 1), "%s.%d.net.1.weight" % (pre, bid + 1))
        put(2 * bid, "%s.%d.net.2.weight" % (pre, bid + 1), "%s.%d.net.2.bias" % (pre, bid + 1), "%s.%d.net.3.weight" % (pre, bid + 1))
    put(11, pre + ".7.weight", pre + ".7.bias", None)
    return ck


def main():
    if not os.path.isdir(REF):
        print("reference tree not mounted: nothing to generate")
        return 0
    map_cuda_to_cpu()
    demo = import_reference_demo()
    from util import latent, make_main_params, make_imp_params
    tmp = tempfile.mkdtemp(prefix="drv_")
    os.makedirs(OUT, exist_ok=True)

    def load(drv, ck, cast):
        drv = drv.to("cuda:0")                                              # as encoding() / decoding() do (lic360_demo.py:350-355)
        drv.load_state_dict(cast(ck, drv.state_dict()))
        return cpu_keys(drv)

    # ---- case A: the demo's own flow at G = 48 on a small ERP: importance map -> its bitstream -> decoded mask -> latent bitstream
    # case c (round 5): 64 rows -- full-lane diagonals of the decode kernels, 64-row windows, corner diagonals that pack two / three samples per task
    only = set(sys.argv[1:])
    for tag, G, H, W, seed in (("a", 48, 8, 12, 4101), ("b", 48, 6, 10, 4102), ("c", 48, 64, 20, 4103)):
        if only and tag not in only:
            continue
        wseed = 1000 + seed
        layers, imp_layers = make_main_params(wseed, G), make_imp_params(wseed)
        code, mask, levels = latent(np.random.default_rng(seed), G, H, W)
        ck, ick = checkpoint_main(layers), checkpoint_imp(imp_layers)
        f_lat, f_imp = os.path.join(tmp, tag), os.path.join(tmp, tag + "_imp")
        imp_enc = load(demo.ImpEntEncoderFast(), ick, demo.cast_imp_entropy_parameter)
        imp_enc.start(f_imp)
        imp_enc.forward(torch.from_numpy(levels))
        enc = load(demo.EntEncoderFast(ngroup=G), ck, demo.cast_entropy_parameter)
        enc.start(f_lat)
        enc.forward(torch.from_numpy(code), torch.from_numpy(mask))
        imp_dec = load(demo.ImpEntDecoder(), ick, demo.cast_imp_entropy_parameter)
        imp_dec.start(f_imp)
        tmask = imp_dec.forward(H // 2, W // 2)
        dec = load(demo.EntDecoder(ngroup=G), ck, demo.cast_entropy_parameter)
        dec.start(f_lat)
        tcode = dec.forward(tmask)
        lat_bytes, imp_bytes = open(f_lat, "rb").read(), open(f_imp, "rb").read()
        assert np.array_equal(tmask.numpy(), mask), "the reference's ImpEntDecoder did not give back the encoder's mask"
        assert np.array_equal(tcode.numpy(), code * mask), "the reference's EntDecoder did not give back code * mask"
        np.savez_compressed(os.path.join(OUT, "driver_%s.npz" % tag), G=G, H=H, W=W, seed=seed, wseed=wseed, code=code, mask=mask, levels=levels,
                            latent_bytes=np.frombuffer(lat_bytes, np.uint8), imp_bytes=np.frombuffer(imp_bytes, np.uint8),
                            decoded_mask=tmask.numpy(), decoded_code=tcode.numpy())
        print("driver_%s: G %d, %dx%d: latent %d B, importance %d B, decoded code == code*mask: %s" %
              (tag, G, H, W, len(lat_bytes), len(imp_bytes), bool(np.array_equal(tcode.numpy(), code * mask))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
