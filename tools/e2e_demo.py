"""Compress and decompress ERP images with this backend alone (no reference Python): analysis transform -> device-resident entropy codecs
-> one container file per image -> decode -> synthesis transform, with bitrate, PSNR, viewport PSNR / SSIM and timings.

    python tools/e2e_demo.py --out /tmp/lic360_demo [--images a.npy b.npy ...] [--checkpoint model.pt --imp-checkpoint imp.pt]

Images are float32 or uint8 arrays [512,1024,3] (.npy); without --images two synthetic ERPs are made.  Without checkpoints the
networks carry seeded random weights (the pipeline runs, the pictures mean nothing); with the reference's checkpoints
(`<prex>_v0_best_0.pt`, `<prex>_imp_best_0.pt`, test/lic360_demo.py:341-344) the transforms load them by name and the entropy models through
the reference's key map (lic360_codec.cast_entropy_parameter / cast_imp_entropy_parameter)."""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def synthetic_erp(seed):
    rng = np.random.default_rng(seed)
    img = rng.random((16, 32, 3)).astype(np.float32)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    return torch.nn.functional.interpolate(t, size=(512, 1024), mode="bicubic", align_corners=False).clamp(0, 1)[0].permute(1, 2, 0).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--images", nargs="*", default=[])
    ap.add_argument("--checkpoint")
    ap.add_argument("--imp-checkpoint")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import lic360_container as box
    import lic360_models as lm
    import lic360_operator as lo
    from lic360_fused import FusedCodec, FusedImpCodec
    dev = "cuda:0"
    imgs = [np.load(f) for f in args.images] or [synthetic_erp(s) for s in (1, 2)]
    imgs = [(i.astype(np.float32) / 255.0 if i.dtype == np.uint8 else i.astype(np.float32)) for i in imgs]
    x = torch.from_numpy(np.stack(imgs)).permute(0, 3, 1, 2).contiguous().to(dev)
    n = x.shape[0]
    enc, dec = lm.CMP_Encoder(gpu_id=0).to(dev).eval(), lm.CMP_Decoder(gpu_id=0).to(dev).eval()
    fc, ic = FusedCodec(48, 64, 128, max_batch=n), FusedImpCodec(32, 64, max_batch=n)
    if args.checkpoint:
        import lic360_codec as drv
        params, pimp = torch.load(args.checkpoint, map_location=dev), torch.load(args.imp_checkpoint, map_location=dev)
        enc.load_state_dict({k: params[k] for k in enc.state_dict()})
        dec.load_state_dict({k: params[k] for k in dec.state_dict()})
        e, ie = drv.EntEncoderFast(48).to(dev), drv.ImpEntEncoderFast().to(dev)
        e.load_state_dict(drv.cast_entropy_parameter(params, e.state_dict()))
        ie.load_state_dict(drv.cast_imp_entropy_parameter(pimp, ie.state_dict()))
        fc.load_from_driver(e)
        ic.load_from_driver(ie)
    else:
        from util import make_main_params, make_imp_params
        torch.manual_seed(0)
        dec.quant.weight.data.copy_(enc.quant.weight.data)
        fc.load_layers(make_main_params(1003, 48))
        ic.load_layers(make_imp_params(1003))
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.time()
        code, mask, levels = enc(x)
        streams, istreams = fc.encode(code.contiguous(), mask.contiguous()), ic.encode(levels.contiguous())
        torch.cuda.synchronize()
        t1 = time.time()
        files = []
        for i in range(n):
            f = os.path.join(args.out, "img%03d.lic360" % i)
            open(f, "wb").write(box.pack(streams[i], istreams[i], 512, 1024, 0, False))
            files.append(f)
        t2 = time.time()
        blobs = [box.unpack(open(f, "rb").read()) for f in files]
        lv = ic.decode([b["imp"] for b in blobs])
        m2 = (torch.arange(48, device=dev).view(1, 48, 1, 1) < lv.repeat_interleave(2, 2).repeat_interleave(2, 3)).float()
        rec = dec(fc.decode([b["latent"] for b in blobs], m2), m2).clamp(0, 1)
        torch.cuda.synchronize()
        t3 = time.time()
        pr = lo.MultiProject(171, 256, 0.5, False, 0)
        va, vb = pr(x).clone(), pr(rec.contiguous())
        ssim = lo.SSIM(11, 3)
    for i in range(n):
        mse = float(torch.mean((x[i] - rec[i]) ** 2))
        vm = float(torch.mean((va[i * 14:(i + 1) * 14] - vb[i * 14:(i + 1) * 14]) ** 2))
        print("%s  %.3f bpp  PSNR %.2f dB  viewport PSNR %.2f dB  viewport SSIM %.4f" % (
            files[i], os.path.getsize(files[i]) * 8 / (512.0 * 1024.0), 10 * math.log10(1.0 / max(mse, 1e-12)), 10 * math.log10(1.0 / max(vm, 1e-12)),
            float(ssim(va[i * 14:(i + 1) * 14], vb[i * 14:(i + 1) * 14]))))
        np.save(os.path.join(args.out, "img%03d_decoded.npy" % i), (rec[i].permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8))
    print("encode %.1f ms, decode %.1f ms for %d image(s) (first call: includes library autotuning)" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, n))


if __name__ == "__main__":
    sys.exit(main())
