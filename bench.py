#!/usr/bin/env python3
"""bench.py -- ERP Mpixels/s enc+dec @512x1024, model-idx 3 (BASELINE.json metric) on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic images per GPU: encode the quantised
latents (code, mask resident in HBM) into bitstreams (HBM) and decode those bitstreams back, through the
device-resident codec (lic360_fused.FusedCodec -> liblic360_hip.so).  Images shard one-per-slot across the
GPUs with no data-path collective: every rank runs the same per-GPU batch ("weak" scaling);
the only cross-rank traffic is the barrier + MAX of the wall time.

Prints ONE JSON line on rank 0 (contract in the task description), with
  roofline:     dominant kernel = the hidden-layer masked conv (encode-order k_cconv_ec or decode-order
                k_cconv_dc, whichever took more time); achieved = algorithmic FLOPs per launch / mean launch
                duration from HIP events recorded on the launch stream during the timed steps;
  cpu_baseline: the CPU oracle (oracle/, OpenMP over output scalars) on a bounded crop of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "360-image-compression_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

G, H, W = 48, 64, 128                   # entropy-domain tensor of a 512x1024 ERP
PIXELS = 512 * 1024
MODEL_IDX, SSIM = 3, 1
# algorithmic work per image, exact counts from the mask rule (SURVEY.md §8d / BASELINE.md §3)
HIDDEN_GMAC = 11.23                     # one hidden layer, 3 stacked nets
F32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def synth_latents(batch, seed0):
    from util import latent
    codes, masks = [], []
    for i in range(batch):
        c, m, _ = latent(np.random.default_rng(seed0 + i), G, H, W)
        codes.append(c)
        masks.append(m)
    return np.concatenate(codes, 0), np.concatenate(masks, 0)


def cpu_baseline(layers):
    """Oracle enc+dec of a 16x32 crop of one latent (1/16 of an image: full 48-group, 12-layer, 3-net model)."""
    import ref_codec as rc
    from util import latent
    ch, cw = 16, 32
    code, mask, _ = latent(np.random.default_rng(99), G, ch, cw)
    t0 = time.time()
    data = rc.encode_main(code, mask, layers, G)
    out = rc.decode_main(data, mask, layers, G)
    dt = time.time() - t0
    assert np.array_equal(out, code * mask)
    px = PIXELS * (ch * cw) / float(H * W)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": px / dt / 1e6, "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": "oracle enc+dec of one %dx%d latent crop (=%d px of a 512x1024 ERP), full 12-layer x3 model, %.1f s" % (ch, cw, int(px), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (BASELINE.json configs[3]: 8 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import ref_codec as rc
    from lic360_fused import FusedCodec
    layers = rc.make_main_params(1000 * SSIM + MODEL_IDX, G)
    B = args.batch
    codec = FusedCodec(G, H, W, max_batch=B, device=local)
    codec.load_layers(layers)
    code_np, mask_np = synth_latents(B, seed0=1000 * rank)
    code, mask = torch.from_numpy(code_np).to(dev), torch.from_numpy(mask_np).to(dev)

    def step():
        codec.encode_async(code, mask)
        codec.decode_async(mask, B)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    # correctness of what is being timed: decode(encode(x)) == x on every rank, no coder faults
    ok = bool(torch.equal(codec.code_out[:B], code * mask)) and int(codec.err[:B].abs().sum().item()) == 0
    codec.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = codec.profile_read()
    codec.profile(False)
    ok = ok and bool(torch.equal(codec.code_out[:B], code * mask))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        okt = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        images = world * B * args.steps
        value = images * PIXELS / dt / 1e6
        nbytes = codec.nbytes[:B].cpu().numpy()
        # dominant kernel: hidden-layer masked conv, encode order vs decode order
        ec_t, dc_t = prof["ec_ms"], prof["dc_ms"]
        if dc_t >= ec_t:
            name, tot_ms, launches = "k_cconv_dc (decode order, hidden layer)", dc_t, prof["dc_launches"]
            flops_per_launch = 2 * HIDDEN_GMAC * 1e9 * B * 10 * args.steps / max(launches, 1)
        else:
            name, tot_ms, launches = "k_cconv_ec (encode order, hidden layer)", ec_t, prof["ec_launches"]
            flops_per_launch = 2 * HIDDEN_GMAC * 1e9 * B
        avg_ms = tot_ms / max(launches, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        out = {
            "metric": "ERP Mpixels/s enc+dec @512x1024 model-idx 3; bitstream bit-exact vs ref",
            "value": value, "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic 512x1024 ERP latents per GPU (48x64x128 symbols + importance mask), "
                                   "model-idx 3 --ssim seeded weights, latent entropy encode+decode (BASELINE.json configs[3] per-GPU share)" % B,
                       "images_per_gpu_per_step": B, "roundtrip_exact": ok, "mean_bitstream_bytes": float(nbytes.mean())},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "avg_launch_ms": avg_ms, "launches": launches,
                         "ec_hidden_ms_per_step": ec_t / args.steps, "dc_hidden_ms_per_step": dc_t / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(layers)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
