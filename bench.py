#!/usr/bin/env python3
"""bench.py -- ERP Mpixels/s enc+dec @512x1024, model-idx 3 (BASELINE.json metric) on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic images per GPU: encode the quantised
latents (code, mask resident in HBM) into bitstreams (HBM) and decode those bitstreams back, through the
device-resident codec (lic360_fused.FusedCodec -> liblic360_hip.so).  Images shard one-per-slot across the
GPUs with no data-path collective: every rank runs the same per-GPU batch ("weak" scaling);
the only cross-rank traffic is the barrier + MAX of the wall time.

Prints ONE JSON line on rank 0 (contract in the task description), with
  roofline:     dominant kernel = the hidden-layer masked conv (encode order or decode order, whichever took more
                time).  Its binding roofline is the fp32 MFMA peak: the MACs of a launch take longer at 157 TFLOP/s
                than its algorithmic bytes (the 9-diagonal halo of every live group, read once) take at 8 TB/s.
                achieved = algorithmic FLOPs per launch / mean launch duration from HIP events recorded on the
                launch stream during the timed steps; the HBM view of the same launches (algorithmic bytes and the
                measured PMC traffic) is reported next to it;
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

# the oracle's OpenMP pool: the GPU box gives one GPU a 16-core CPU share
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

G, H, W = 48, 64, 128                   # entropy-domain tensor of a 512x1024 ERP
PIXELS = 512 * 1024
MODEL_IDX, SSIM = 3, 1
# algorithmic work per image, exact counts from the mask rule (SURVEY.md §8d / BASELINE.md §3)
HIDDEN_GMAC = 11.23                     # one hidden layer, 3 stacked nets
NET_GMAC = 123.42                       # the whole 12-layer x 3-net latent entropy model, one direction
F32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 MFMA dense peak
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def chain_len(g, hidden=1):
    """input groups group g reads in a hidden layer (extension/cconv_ec_cuda.cu:288-290, longest lane)"""
    return min(G, g + 4 + hidden)


def dc_hidden_bytes_per_launch(b):
    """Algorithmic HBM bytes of one decode-order hidden-layer launch (one anti-diagonal plane, 3*b sample-nets), averaged
    over the planes: every input (channel, diagonal) a live group needs is read once, every output written once, half of
    the hidden layers read a residual, the weights of the live groups are read once."""
    S, n = H + W - 1, 3 * b
    lens = [min(s, H - 1) - max(0, s - W + 1) + 1 for s in range(S)]
    tot, launches = 0.0, 0
    for p in range(S + G - 1):
        live = [g for g in range(G) if 0 <= p - g < S]
        if not live:
            continue
        launches += 1
        inb = 0
        for tc in range(G):
            rows = set()
            for g in live:
                if tc < chain_len(g):
                    rows.update(s for s in range(p - g - 4, p - g + 5) if 0 <= s < S)
            inb += 4 * sum(lens[s] for s in rows)
        outb = sum(4 * lens[p - g] for g in live)
        wb = 3 * sum(chain_len(g) * 400 for g in live)
        tot += 4.0 * (n * (inb + 1.5 * outb) + wb)
    return tot / launches


def ec_hidden_bytes_per_launch(b):
    """Algorithmic HBM bytes of one encode-order hidden-layer launch: activations in, activations out (+ residual for
    half of the layers), weights once."""
    n = 3 * b
    return 4.0 * (n * 4 * G * H * W * 2.5 + 3 * sum(chain_len(g) * 400 for g in range(G)))


def synth_latents(batch, seed0):
    from util import latent
    codes, masks, levels = [], [], []
    for i in range(batch):
        c, m, lv = latent(np.random.default_rng(seed0 + i), G, H, W)
        codes.append(c)
        masks.append(m)
        levels.append(lv)
    return np.concatenate(codes, 0), np.concatenate(masks, 0), np.concatenate(levels, 0)


def cpu_baseline(layers):
    """Oracle enc+dec of four 32x32 latent crops (together half an image's symbols: full 48-group, 12-layer, 3-net model)."""
    import ref_codec as rc
    from util import latent
    ch, cw, reps = 32, 32, 4
    t0 = time.time()
    for i in range(reps):
        code, mask, _ = latent(np.random.default_rng(99 + i), G, ch, cw)
        data = rc.encode_main(code, mask, layers, G)
        out = rc.decode_main(data, mask, layers, G)
        assert np.array_equal(out, code * mask)
    dt = time.time() - t0
    px = reps * PIXELS * (ch * cw) / float(H * W)
    cores = int(os.environ["OMP_NUM_THREADS"])
    # single-thread arithmetic coder alone: config-1 workload, 393 216 symbols on per-symbol 9-entry tables
    import oracle as orc
    rng = np.random.default_rng(5)
    nsym = G * H * W
    inner = np.sort(rng.integers(1, 65535, (nsym, 7)), axis=1) + np.arange(7)
    tab = np.concatenate([np.zeros((nsym, 1), np.int64), inner, np.full((nsym, 1), 65536 + 7, np.int64)], 1).astype(np.int32)
    sym = rng.integers(0, 8, nsym).astype(np.int32)
    t1 = time.time()
    enc = orc.Encoder()
    enc.encode(tab, 8, sym, None, nsym)
    data = enc.finish()
    t2 = time.time()
    dec = orc.Decoder(data)
    out = dec.decode(tab, 8, None, nsym)
    dec.close()
    t3 = time.time()
    assert np.array_equal(out.astype(np.int32), sym)
    return {"value": px / dt / 1e6, "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "coder_single_thread_Msym_per_s": {"encode": nsym / (t2 - t1) / 1e6, "decode": nsym / (t3 - t2) / 1e6},
            "sample": "oracle enc+dec of %d latent crops of %dx%d (=%d px of 512x1024 ERPs), full 12-layer x3 model, %.1f s" % (reps, ch, cw, int(px), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=144, help="images per GPU per step (BASELINE.json configs[3] shards a 64-image batch; "
                    "144 resident images per GPU (3 sub-batches of 48, 52 GB of the 288 GB) keep the decode wavefront wide enough to fill "
                    "256 CUs: 96 -> 42.2, 144 -> 43.9, 192 -> 44.3 Mpixel/s)")
    ap.add_argument("--streams", type=int, default=3, help="the per-GPU batch is split over this many HIP streams so that one "
                    "sub-batch's serial arithmetic-coder phases overlap the other's convolutions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-importance-pass", action="store_true", help="skip the extra (untimed for `value`) pass that also codes the importance maps")
    args = ap.parse_args()

    import lic360_shard as shard
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    rank, local, world = shard.init_from_env("nccl")          # one process per GPU; "nccl" is RCCL on ROCm
    dev = torch.device("cuda", local)

    from util import make_main_params, make_imp_params          # seeded synthetic weights (numpy only)
    from lic360_fused import FusedCodec
    layers = make_main_params(1000 * SSIM + MODEL_IDX, G)
    B = args.batch
    ns = max(1, min(args.streams, B))
    sizes = [B // ns + (1 if i < B % ns else 0) for i in range(ns)]
    code_np, mask_np, level_np = synth_latents(B, seed0=1000 * rank)
    codecs, codes, masks, streams = [], [], [], []
    o = 0
    for sz in sizes:
        c = FusedCodec(G, H, W, max_batch=sz, device=local)
        c.load_layers(layers)
        codecs.append(c)
        codes.append(torch.from_numpy(code_np[o:o + sz]).to(dev))
        masks.append(torch.from_numpy(mask_np[o:o + sz]).to(dev))
        streams.append(torch.cuda.Stream(device=dev))
        o += sz
    torch.cuda.synchronize(dev)

    def step():
        for c, cd, mk, st in zip(codecs, codes, masks, streams):
            with torch.cuda.stream(st):
                c.encode_async(cd, mk)
        for c, cd, mk, st in zip(codecs, codes, masks, streams):
            with torch.cuda.stream(st):
                c.decode_async(mk, cd.shape[0])

    def roundtrip_ok():
        return all(bool(torch.equal(c.code_out[:cd.shape[0]], cd * mk)) and int(c.err[:cd.shape[0]].abs().sum().item()) == 0
                   for c, cd, mk in zip(codecs, codes, masks))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    # correctness of what is being timed: decode(encode(x)) == x on every rank, no coder faults
    ok = roundtrip_ok()
    for c in codecs:
        c.profile(True)
    dt = shard.timed(step, args.steps, dev)                   # barrier + sync both sides, MAX over ranks
    prof = {"ec_ms": 0.0, "ec_launches": 0, "dc_ms": 0.0, "dc_launches": 0}
    for c in codecs:
        pr = c.profile_read()
        for k in prof:
            prof[k] += pr[k]
        c.profile(False)
    ok = ok and roundtrip_ok()
    # the same kernels with the GPU to themselves: one sub-batch, one stream (per-launch durations above are stretched by
    # the other streams' kernels sharing the CUs)
    codecs[0].profile(True)
    with torch.cuda.stream(streams[0]):
        codecs[0].encode_async(codes[0], masks[0])
        codecs[0].decode_async(masks[0], codes[0].shape[0])
    torch.cuda.synchronize(dev)
    iso = codecs[0].profile_read()
    codecs[0].profile(False)
    # extra pass, reported next to `value`: the same steps with the images' importance-map streams (32x64 maps, 49 levels,
    # device-resident FusedImpCodec on the same streams) encoded and decoded as well -- both bitstreams of every image
    with_imp = None
    if not args.no_importance_pass:
        from lic360_fused import FusedImpCodec
        imp_layers = make_imp_params(1000 * SSIM + MODEL_IDX)
        icodecs, levels, o = [], [], 0
        for sz in sizes:
            ic = FusedImpCodec(H // 2, W // 2, max_batch=sz, device=local)
            ic.load_layers(imp_layers)
            icodecs.append(ic)
            levels.append(torch.from_numpy(level_np[o:o + sz]).to(dev))
            o += sz

        def step_full():
            for c, ic, cd, mk, lv, st in zip(codecs, icodecs, codes, masks, levels, streams):
                with torch.cuda.stream(st):
                    ic.encode_async(lv)
                    c.encode_async(cd, mk)
            for c, ic, cd, mk, lv, st in zip(codecs, icodecs, codes, masks, levels, streams):
                with torch.cuda.stream(st):
                    ic.decode_async(lv.shape[0])
                    c.decode_async(mk, cd.shape[0])

        step_full()
        dt_full = shard.timed(step_full, args.steps, dev)
        ok_imp = all(bool(torch.equal(ic.levels_out[:lv.shape[0]], lv)) and int(ic.err[:lv.shape[0]].abs().sum().item()) == 0
                     for ic, lv in zip(icodecs, levels))
        with_imp = {"value": world * B * args.steps * PIXELS / dt_full / 1e6, "unit": "Mpixel/s", "ms_per_step": dt_full / args.steps * 1e3,
                    "roundtrip_exact": shard.all_ok(ok_imp and roundtrip_ok(), dev),
                    "mean_importance_bytes": float(np.mean([float(ic.nbytes[:lv.shape[0]].float().mean().item()) for ic, lv in zip(icodecs, levels)]))}
    ok = shard.all_ok(ok, dev)

    if rank == 0:
        images = world * B * args.steps
        value = images * PIXELS / dt / 1e6
        nbytes = np.concatenate([c.nbytes[:cd.shape[0]].cpu().numpy() for c, cd in zip(codecs, codes)])
        # dominant kernel: hidden-layer masked conv, encode order vs decode order
        ec_t, dc_t = prof["ec_ms"], prof["dc_ms"]
        b0 = int(codes[0].shape[0])                        # images per launch (one sub-batch)
        dom_dc = dc_t >= ec_t
        if dom_dc:
            name, tot_ms, launches = "k_cconv4v6<4, false, false> (decode order, hidden layers)", dc_t, prof["dc_launches"]
            iso_ms = iso["dc_ms"] / max(iso["dc_launches"], 1)
            bytes_per_launch = dc_hidden_bytes_per_launch(b0)
        else:
            name, tot_ms, launches = "k_cconv4v6<4, true, false> (encode order, hidden layers)", ec_t, prof["ec_launches"]
            iso_ms = iso["ec_ms"] / max(iso["ec_launches"], 1)
            bytes_per_launch = ec_hidden_bytes_per_launch(b0)
        flops_per_launch = 2 * HIDDEN_GMAC * 1e9 * B * 10 * args.steps / max(launches, 1)   # all hidden launches of a step cover B images x 10 layers
        avg_ms = tot_ms / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        iso_gbs = bytes_per_launch / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
        tf = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        iso_tf = flops_per_launch / (iso_ms * 1e-3) / 1e12 if iso_ms > 0 else 0.0
        traffic = None
        try:      # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/, separate rocprofv3 --pmc runs)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["dominant_kernel"]
            if dom_dc:
                traffic = pm["traffic_bytes_per_launch"] * b0 / pm["images_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "ERP Mpixels/s enc+dec @512x1024 model-idx 3; bitstream bit-exact vs ref",
            "value": value, "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic 512x1024 ERP latents per GPU (48x64x128 symbols + importance mask), "
                                   "model-idx 3 --ssim seeded weights, latent entropy encode+decode (BASELINE.json configs[3] per-GPU share)" % B,
                       "images_per_gpu_per_step": B, "streams": ns, "roundtrip_exact": ok, "mean_bitstream_bytes": float(nbytes.mean()),
                       "with_importance_stream": with_imp},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "algorithmic_flops_per_launch": flops_per_launch, "images_per_launch": b0,
                         "avg_launch_ms": avg_ms, "launches": launches, "concurrent_streams": ns,
                         "isolated": {"achieved": iso_tf, "frac": iso_tf / F32_MFMA_PEAK_TFLOPS, "avg_launch_ms": iso_ms,
                                      "note": "same kernel, one sub-batch alone on the GPU (single stream); in the timed run "
                                              "kernels of the other streams share the CUs and stretch each launch"},
                         "hbm": {"algorithmic_bytes_per_launch": bytes_per_launch, "achieved": achieved, "isolated": iso_gbs,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "frac_isolated": iso_gbs / HBM_PEAK_GBS},
                         "aggregate": {"achieved": 2 * 2 * NET_GMAC * 1e9 * B / (dt / args.steps) / 1e12, "unit": "TFLOP/s",
                                       "frac": 2 * 2 * NET_GMAC * 1e9 * B / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                       "note": "algorithmic FLOPs of every conv layer of encode + decode of the step / wall time of the step "
                                               "(all streams; includes the arithmetic-coder phases that are not hidden)"},
                         "ec_hidden_ms_per_step": ec_t / args.steps, "dc_hidden_ms_per_step": dc_t / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(layers)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
