"""Analysis / synthesis transforms (lic360_models.py) at the reference's width (192 channels, 512x1024 ERPs): ms per image and nominal
TFLOP/s (library convolutions through MIOpen + this package's sphere / shuffle / quantiser / GDN kernels), and the one-pass GDN against
its four-kernel torch form.  Seeded random weights.  Writes one JSON document to stdout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "360-image-compression_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

F32_PEAK_TFLOPS = 157.3


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def measure(batch=8, device=0, reps=3):
    import lic360
    import lic360_models as lm
    dev = "cuda:%d" % device
    torch.manual_seed(0)
    enc, dec = lm.CMP_Encoder(gpu_id=device).to(dev).eval(), lm.CMP_Decoder(gpu_id=device).to(dev).eval()
    ge, gd = lm.transform_gflops()
    rows = []
    with torch.no_grad():
        img = torch.rand((batch, 3, 512, 1024), device=dev)
        code, mask, _ = enc(img)
        te = timed(lambda: enc(img), reps)
        td = timed(lambda: dec(code, mask), reps)
        for name, t, gf in (("analysis transform (image -> symbols, mask, importance map)", te, ge), ("synthesis transform (symbols, mask -> image)", td, gd)):
            rows.append({"kernel": name, "bound": "mfma", "images_per_launch": batch, "ms_per_image": t / batch * 1e3, "nominal_gflop_per_image": gf,
                         "achieved": gf * batch / t / 1e3, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s (nominal conv + GDN flops)",
                         "frac": gf * batch / t / 1e3 / F32_PEAK_TFLOPS, "how": "torch conv2d (MIOpen) + native sphere / shuffle / quantiser / GDN kernels"})
        # the one-pass GDN on the largest map (260 x 516 x 192) against its torch form
        c = 192
        x = torch.randn((batch, c, 260, 516), device=dev)
        gamma, beta = torch.rand((c, c), device=dev) * 0.02 + 0.1 * torch.eye(c, device=dev), torch.rand((c,), device=dev) + 0.5
        out = torch.empty_like(x)
        t_f = timed(lambda: lic360.gdn_forward(x, gamma, beta, False, out), 5)
        t_t = timed(lambda: x / torch.sqrt(F.conv2d(x * x, gamma.view(c, c, 1, 1), beta)), 5)
        fl = 2.0 * c * c * x.numel() / c
        rows.append({"kernel": "gdn one pass (192 ch, 260x516)", "bound": "mfma", "images_per_launch": batch, "avg_launch_ms": t_f * 1e3,
                     "achieved": fl / t_f / 1e12, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / t_f / 1e12 / F32_PEAK_TFLOPS,
                     "algorithmic_bytes_per_launch": 2.0 * x.numel() * 4, "GBps": 2.0 * x.numel() * 4 / t_f / 1e9,
                     "torch_four_kernel_form_ms": t_t * 1e3, "speedup_vs_torch": t_t / t_f})
        # the hand-written 3x3 convolution (csrc/conv3x3_kernels.hip) on the two shapes that carry the transforms, against the library convolution alone
        # and against the library form of everything the kernel does (in-place apron refresh + conv + PReLU + trim + residual add)
        for hp, wp in ((260, 516), (132, 260)):
            xx = torch.randn((batch, c, hp, wp), device=dev)
            ww, bb, sl = torch.randn((c, c, 3, 3), device=dev) * 0.05, torch.randn((c,), device=dev), torch.rand((c,), device=dev) * 0.5
            res, oo, pk = torch.randn_like(xx), torch.zeros_like(xx), lic360.sconv3x3_pack(ww)
            pad_op, trim_op = lic360.SpherePadOp(2, True, device, False), lic360.SphereTrimOp(2, device, False)
            t_o = timed(lambda: lic360.sconv3x3(xx, pk, bb, sl, res, oo, pad=2, sphere=True, ring=2), 5)
            t_c = timed(lambda: F.conv2d(xx, ww, bb, padding=1), 5)
            t_l = timed(lambda: trim_op.forward(F.prelu(F.conv2d(pad_op.forward(xx)[0], ww, bb, padding=1), sl))[0] + res, 5)
            fl = 2.0 * batch * c * c * 9 * (hp - 4) * (wp - 4)
            rows.append({"kernel": "sconv3x3 192->192 on %dx%d maps (%dx%d window): apron by index + conv + bias + PReLU + trim + residual" % (hp, wp, hp - 4, wp - 4),
                         "bound": "mfma", "images_per_launch": batch, "avg_launch_ms": t_o * 1e3, "achieved": fl / t_o / 1e12, "peak": F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": fl / t_o / 1e12 / F32_PEAK_TFLOPS, "algorithmic_flops_per_launch": fl,
                         "miopen_conv_alone_ms": t_c * 1e3, "miopen_conv_with_pad_prelu_trim_add_ms": t_l * 1e3,
                         "speedup_vs_miopen_conv_alone": t_c / t_o, "speedup_vs_library_form": t_l / t_o})
    one = whole_codec(enc, dec, device)                                     # 48 images, one stream (+ the importance codec on a side stream)
    row = whole_codec_streams(enc, dec, device, batch=192, nstreams=3)      # the bench's own batch structure: 192 images = 3 sub-batches of 64 on 3 streams (round 6; 144 = 3 x 48 before)
    row["one_stream_48_images_mpixel_s"], row["one_stream_48_images_ms_per_image"] = one["achieved"], one["ms_per_image"]
    row["roundtrip_exact"] = bool(row["roundtrip_exact"] and one["roundtrip_exact"])
    rows.append(row)
    return rows


def whole_codec(enc, dec, device, batch=48, reps=2):
    """image -> analysis -> both entropy encoders -> bitstreams (in HBM) -> both entropy decoders -> synthesis -> image, one stream, seeded
    weights: the end-to-end rate of the codec on one GPU, one batch of 48 on one stream (see whole_codec_streams for the bench's own structure)."""
    import time
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_main_params, make_imp_params
    from lic360_fused import FusedCodec, FusedImpCodec
    dev = "cuda:%d" % device
    G = 48
    fc, ic = FusedCodec(G, 64, 128, max_batch=batch, device=device), FusedImpCodec(32, 64, max_batch=batch, device=device)
    fc.load_layers(make_main_params(1003, G))
    ic.load_layers(make_imp_params(1003))
    lvl = torch.arange(G, device=dev).view(1, G, 1, 1)
    with torch.no_grad():
        dec.quant.weight.copy_(enc.quant.weight)
        img = torch.rand((batch, 3, 512, 1024), device=dev)
        chunks = [slice(i, i + 8) for i in range(0, batch, 8)]

        side = torch.cuda.Stream(device=dev)                                # the importance-map stream of the batch (as in bench.py: a stream of its own)

        def run():
            main = torch.cuda.current_stream()
            parts = [enc(img[c]) for c in chunks]                           # the transforms run in sub-batches of 8 (activation memory)
            code, mask, lv = (torch.cat([p[k] for p in parts]).contiguous() for k in range(3))
            side.wait_stream(main)
            with torch.cuda.stream(side):                                   # the map's encode + decode (38 ms of small launches) under the latent's encode (136 ms)
                ic.encode_async(lv)
                ic.decode_async(batch)
                lv2 = ic.levels_out[:batch]
                mask2 = (lvl < lv2.repeat_interleave(2, 2).repeat_interleave(2, 3)).float()
            lv.record_stream(side)
            mask2.record_stream(main)
            fc.encode_async(code, mask)
            main.wait_stream(side)                                          # the latent decode needs the mask the decoded map gives
            fc.decode_async(mask2, batch)
            code2 = fc.code_out[:batch]
            rec = torch.cat([dec(code2[c], mask2[c]) for c in chunks])
            return code, mask, code2, rec
        code, mask, code2, rec = run()
        torch.cuda.synchronize()
        exact = bool(torch.equal(code2, code * mask)) and bool(torch.isfinite(rec).all())
        t0 = time.time()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
    return {"kernel": "whole codec: analysis + entropy encode + entropy decode + synthesis", "bound": "mfma", "images_per_launch": batch,
            "ms_per_image": dt / batch * 1e3, "achieved": batch * 512 * 1024 / dt / 1e6, "unit": "Mpixel/s (one stream + the importance-map codec on a side stream, both directions, transforms included)",
            "peak": None, "frac": None, "roundtrip_exact": exact, "mean_latent_bytes": float(fc.nbytes[:batch].float().mean().item())}


def whole_codec_streams(enc, dec, device, batch=48, reps=2, nstreams=2):
    """image -> analysis -> both entropy encoders -> bitstreams (in HBM) -> both entropy decoders -> synthesis -> image, seeded weights: the
    end-to-end rate of the codec on one GPU.  The batch runs as `nstreams` independent sub-batches on HIP streams of their own (as bench.py runs
    the entropy path): one sub-batch's launch-bound decode planes and serial coder chains fill under another's transforms.  Every stream has its
    own copy of the networks (the fused blocks keep per-module work buffers) and its own codecs.  Measured (round 5): 48 images per stream on
    1 / 2 / 3 streams 24.5 / 25.4 / 25.7 Mpixel/s; 48 images SPLIT over 2 / 3 streams 22.8 / 20.0 (the coder batches lose their sample packing)."""
    import copy
    import time
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_main_params, make_imp_params
    from lic360_fused import FusedCodec, FusedImpCodec
    dev = "cuda:%d" % device
    G = 48
    per = batch // nstreams
    main_p, imp_p = make_main_params(1003, G), make_imp_params(1003)
    lvl = torch.arange(G, device=dev).view(1, G, 1, 1)
    with torch.no_grad():
        dec.quant.weight.copy_(enc.quant.weight)
        lanes = []
        for i in range(nstreams):
            fc, ic = FusedCodec(G, 64, 128, max_batch=per, device=device), FusedImpCodec(32, 64, max_batch=per, device=device)
            fc.load_layers(main_p)
            ic.load_layers(imp_p)
            lanes.append({"enc": enc if i == 0 else copy.deepcopy(enc), "dec": dec if i == 0 else copy.deepcopy(dec), "fc": fc, "ic": ic,
                          "st": torch.cuda.Stream(device=dev), "img": torch.rand((per, 3, 512, 1024), device=dev)})
        torch.cuda.synchronize()

        def run():
            outs = []
            for L in lanes:
                with torch.cuda.stream(L["st"]):
                    chunks = [slice(i, min(i + 8, per)) for i in range(0, per, 8)]   # the transforms run in chunks of 8 (activation memory)
                    parts = [L["enc"](L["img"][c]) for c in chunks]
                    code, mask, lv = (torch.cat([p[k] for p in parts]).contiguous() for k in range(3))
                    L["fc"].encode_async(code, mask)
                    L["ic"].encode_async(lv)
                    L["ic"].decode_async(per)
                    lv2 = L["ic"].levels_out[:per]
                    mask2 = (lvl < lv2.repeat_interleave(2, 2).repeat_interleave(2, 3)).float()
                    L["fc"].decode_async(mask2, per)
                    code2 = L["fc"].code_out[:per]
                    rec = torch.cat([L["dec"](code2[c], mask2[c]) for c in chunks])
                    outs.append((code, mask, code2, rec))
            torch.cuda.synchronize()
            return outs
        outs = run()
        exact = all(bool(torch.equal(c2, c * m)) and bool(torch.isfinite(r).all()) for c, m, c2, r in outs)
        t0 = time.time()
        for _ in range(reps):
            run()
        dt = (time.time() - t0) / reps
    n = per * nstreams
    return {"kernel": "whole codec: analysis + entropy encode + entropy decode + synthesis", "bound": "mfma", "images_per_launch": n, "streams": nstreams,
            "ms_per_image": dt / n * 1e3, "achieved": n * 512 * 1024 / dt / 1e6, "unit": "Mpixel/s (%d streams, both directions, transforms included)" % nstreams,
            "peak": None, "frac": None, "roundtrip_exact": exact, "mean_latent_bytes": float(np.mean([float(L["fc"].nbytes[:per].float().mean().item()) for L in lanes]))}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "streams":                      # the whole codec as N sub-batches on N streams (experiment: N = 2, 3)
        import lic360_models as lm
        torch.manual_seed(0)
        e, d = lm.CMP_Encoder(gpu_id=0).to("cuda:0").eval(), lm.CMP_Decoder(gpu_id=0).to("cuda:0").eval()
        per = int(os.environ.get("WC_PER", 0))                              # images per stream (default: 48 in all, split over the streams)
        for ns in (int(v) for v in sys.argv[2:]):
            print(json.dumps(whole_codec_streams(e, d, 0, batch=per * ns if per else 48, nstreams=ns)))
    else:
        print(json.dumps({"rows": measure()}, indent=1))
