#!/usr/bin/env python3
"""bench.py -- ERP Mpixels/s enc+dec @512x1024, model-idx 3 (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic images per GPU: BOTH bitstreams of every image -- the
quantised latent (48x64x128 symbols under its importance mask) and the 32x64 importance map, what the reference's
encoding()/decoding() code (test/lic360_demo.py:357-366, 394-402) -- are encoded into HBM-resident bitstreams and decoded
back through the device-resident codecs (lic360_fused -> liblic360_hip.so).  Images shard one-per-slot across the GPUs
with no data-path collective: every rank runs the same per-GPU batch ("weak" scaling); the only cross-rank
traffic is the barrier + MAX of the wall time.

Launch: with `--gpus N > 1` and no WORLD_SIZE in the environment, this process makes NO HIP call and starts N ranks as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`
(lic360_shard.launch_ranks), which is also how the driver launches it; every rank asserts WORLD_SIZE == --gpus.

Rank 0 prints ONE JSON line (contract in the task description) with, besides the headline:
  roofline      the dominant kernel (the conv kernel class with the most GPU time), measured ALONE on the GPU: one sub-batch,
                one stream, HIP events around every launch on the launch stream (an instrumented pass outside the timed
                region; the timed region runs with instrumentation off).  `concurrent` repeats it with all streams running.
  kernels       one row per kernel class of the codec (conv first/hidden/last in both orders against the fp32 MFMA peak,
                table builds against the HBM peak, the serial coder kernels as ns per symbol) + the streaming ops (GB/s)
  single_image  BASELINE.json configs[1]/[2]: latency of one image alone (encode, decode)
  config4       BASELINE.json configs[3]: a fixed list of 64 images sharded i -> rank i mod N (strong scaling figure)
  config4_per_gpu_share  (N = 1 only) the 8 images one rank of configs[3] handles at 8 GPUs, and the 8-GPU figure they predict
  cpu_baseline  rank 0, N = 1: the CPU oracle on ONE WHOLE image of the same workload (encode + decode); the same leg
                asserts that the GPU's bitstream of that image equals the oracle's byte for byte.
"""
import argparse
import json
import os
import sys
import time

T_START = time.time()

# Six HIP streams per rank (three sub-batches x {latent codec, importance-map codec}).  The HIP runtime multiplexes streams over 4
# hardware queues by default: an importance stream that shares a queue with a latent stream runs behind that stream's long kernels,
# and the latent decodes gated on it (lic360_codec_decode_gated) stall -- 49.3 instead of 50.6 Mpixel/s.  Must be set before the
# runtime initialises; inherited by the ranks the supervisor spawns.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "360-image-compression_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

G, H, W = 48, 64, 128                   # entropy-domain tensor of a 512x1024 ERP
NSYM = G * H * W
PLANES = H + W + G - 2
PIXELS = 512 * 1024
MODEL_IDX, SSIM = 3, 1
# algorithmic work per image and direction, exact counts from the mask rule: GMAC of the 3 stacked nets
GMAC = {"first": 2.70, "hidden": 11.23 * 10, "last": 8.42}
NET_GMAC = 123.42
IMP_GMAC = 6.22
F32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 MFMA dense peak
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=192, help="images per GPU per step: 192 resident images = 3 sub-batches of 64 (rounds 1-5: 144 = 3 x 48; "
                    "side figure `batch144_mpixel_s`).  64 images per sub-batch are 8 per (net, XCD) list -- exactly one packing chunk of the dead-cone task "
                    "lists (csrc/need.h); measured 144 / 192 / 240 / 288 / 384: 61.3 / 62.7 / 62.3 / 62.3 / 62.7 Mpixel/s on one box.  "
                    "(BASELINE.json configs[3] is the 8-per-GPU form: see `config4`)")
    ap.add_argument("--streams", type=int, default=3, help="the per-GPU batch is split over this many HIP streams so that one "
                    "sub-batch's serial arithmetic-coder phases overlap the other's convolutions")
    ap.add_argument("--imp-streams", type=int, default=1, help="1: the importance-map codecs run on HIP streams of their own; 0: on their sub-batch's stream")
    ap.add_argument("--masks", choices=("smooth", "iid"), default="smooth", help="importance maps of the synthetic latents: smooth = SURVEY.md 8d's "
                    "L = clip(round(24 + 12 cos(lat) n)), n smooth noise (tests/util.py:latent_smooth; the workload since round 6); iid = every map cell drawn "
                    "independently (rounds 1-5; the adversarial case of the dead-cone skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (skip latent-only pass, single image, config4, streaming ops)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the launch / sharding / reporting path on gloo: no codec work, value 0")
    return ap.parse_args()


def split_for_streams(n, ns):
    """sizes of the sub-batches of n images on ns streams: multiples of 16 where the list is long enough (the decode kernel packs the samples of an
    XCD's list -- tapes / dead-cone task records -- when 8 divides the images per net; 16 images make lists of two), the remainder on the last stream; a list
    shorter than 16 images stays on one stream (it is latency-bound: the per-plane chain is the same for 1 or 8 images)"""
    exp = os.environ.get("LIC360_SPLIT_SMALL")                           # experiment: "4,4,0" = how a list shorter than 16 images is split
    if n < 16 and exp:
        sz = [int(v) for v in exp.split(",")][:ns]
        if sum(sz) == n:
            return sz + [0] * (ns - len(sz))
    if n < 16:
        return [n] + [0] * (ns - 1)
    blocks, rem = divmod(n, 16)
    use = min(ns, blocks)
    sizes = [16 * (blocks // use + (1 if i < blocks % use else 0)) for i in range(use)] + [0] * (ns - use)
    sizes[use - 1] += rem
    return sizes


MASKS = "smooth"


def synth_latents(batch, seed0, h=H, w=W, kind=None):
    import numpy as np
    from util import make_latent
    codes, masks, levels = [], [], []
    for i in range(batch):
        c, m, lv = make_latent(kind or MASKS, np.random.default_rng(seed0 + i), G, h, w)
        codes.append(c)
        masks.append(m)
        levels.append(lv)
    return np.concatenate(codes, 0), np.concatenate(masks, 0), np.concatenate(levels, 0)


def cpu_baseline(layers, imp_layers, code, mask, levels, gpu_bytes, gpu_imp_bytes):
    """The CPU oracle (oracle/, test infrastructure) on ONE WHOLE image: encode + decode of both streams, OpenMP over the (sample, output
    channel) planes of each layer, on this GPU's share of the box's host cores (16 threads: `value`, `cores`).  Beside it the ENCODE direction
    alone on ALL host cores (`all_cores`): the box is shared between its GPUs' jobs, and the decode direction -- 8 568 small
    parallel regions, one per layer and plane -- does not scale on cores other jobs are using (round 4: the whole leg took 75 s on 256 threads
    against 35 s on 16).  Also the checker of the timed data: the GPU's bitstreams of this image must equal the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import ref_codec as rc
    import oracle as orc
    nproc, cpu_model = host_cpu()
    cores = int(orc.lib.orc_set_num_threads(min(16, nproc)))
    progress("cpu_baseline: oracle encode + decode of one image on %d threads" % cores)
    t0 = time.time()
    data = rc.encode_main(code, mask, layers, G)
    imp = rc.encode_imp(levels, imp_layers)
    t1 = time.time()
    progress("cpu_baseline: encode %.1f s; decode ..." % (t1 - t0))
    out = rc.decode_main(data, mask, layers, G)
    lv = rc.decode_imp(imp, imp_layers, H // 2, W // 2)
    t2 = time.time()
    assert np.array_equal(out, code * mask) and np.array_equal(lv, levels)
    same = bool(data == gpu_bytes and imp == gpu_imp_bytes)
    assert same, "GPU bitstream of image 0 differs from the oracle's"
    all_cores = None
    if nproc > cores:
        n_all = int(orc.lib.orc_set_num_threads(nproc))
        progress("cpu_baseline: oracle encode on all %d host threads" % n_all)
        a0 = time.time()
        same_all = rc.encode_main(code, mask, layers, G) == data and rc.encode_imp(levels, imp_layers) == imp
        a1 = time.time()
        orc.lib.orc_set_num_threads(cores)
        all_cores = {"cores": n_all, "encode_s": a1 - a0, "encode_speedup_vs_%d_threads" % cores: (t1 - t0) / (a1 - a0), "bytes_equal": bool(same_all),
                     "note": "encode direction only (36 + 12 layer-sized parallel loops); the box's cores are shared with other jobs"}
    # BASELINE.json configs[0] beside it: the range coder alone, one thread, 393 216 symbols (= 32 x 64 x 192) on the fixed 9-entry CDF
    rng = np.random.default_rng(1234)
    cdf = np.array([0, 1200, 5200, 14000, 32768, 51536, 60336, 64336, 65536], np.int32)
    sym = np.searchsorted(cdf, rng.integers(0, 65536, 393216), side="right").astype(np.int32) - 1
    tabs = np.tile(cdf, (sym.size, 1))
    c0 = time.time()
    e = orc.Encoder()
    e.encode(tabs, 8, sym, None, sym.size)
    blob = e.finish()
    c1 = time.time()
    d = orc.Decoder(blob)
    back = d.decode(tabs, 8, None, sym.size)
    d.close()
    c2 = time.time()
    assert np.array_equal(back.astype(np.int32), sym)
    return {"value": PIXELS / (t2 - t0) / 1e6, "unit": "Mpixel/s", "cores": cores, "kind": "port", "nproc": nproc, "cpu_model": cpu_model,
            "coder_single_thread": {"symbols": int(sym.size), "bytes": len(blob), "encode_Msym_per_s": sym.size / (c1 - c0) / 1e6,
                                    "decode_Msym_per_s": sym.size / (c2 - c1) / 1e6, "note": "configs[0]: range coder only, fixed CDF, one host thread"},
            "encode_s": t1 - t0, "decode_s": t2 - t1, "gpu_bytes_equal_oracle_bytes": same,
            "all_cores": all_cores,
            "sample": "oracle encode + decode of ONE whole 512x1024 image of the timed batch (image 0: latent 48x64x128 + 32x64 importance "
                      "map, full 12-layer x3 model), %.1f s on %d threads" % (t2 - t0, cores)}


def progress(msg):
    """one line on stderr per stage (a GPU box kills a command that writes nothing for seven minutes)"""
    sys.stderr.write("[bench %6.1f s] %s\n" % (time.time() - T_START, msg))
    sys.stderr.flush()


def host_cpu():
    """(host cores this process may run on, CPU model string) -- SURVEY.md 8d asks for both beside the CPU figure"""
    try:
        nproc = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc = os.cpu_count() or 1
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return nproc, model


def calibrate(dev):
    """Two figures of THIS box, measured in this process before the timed region (~1 s): the fp32 MFMA rate (lic360_calib_mfma_f32: v_mfma_f32_16x16x4_f32
    on every CU) and a 1 GiB device-to-device copy -- plus the card's current shader / memory clocks from sysfs where readable.  The boxes of the pool
    differ by 1-3 %: value / calib_mfma_tflops is what to compare between rounds."""
    import ctypes as C
    import glob
    import torch
    from lic360 import _lib, _chk
    out = {}
    tf, cus = C.c_double(0.0), C.c_int(0)
    _chk(_lib.lic360_calib_mfma_f32(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.byref(tf), C.byref(cus)))
    out["calib_mfma_tflops"], out["compute_units"] = tf.value, cus.value
    n = 1 << 30
    a, b = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    out["calib_copy_gbs"] = 4 * 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9        # bytes read + bytes written
    del a, b
    torch.cuda.empty_cache()
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        try:
            for path in sorted(glob.glob("/sys/class/drm/card*/device/" + name)):
                cur = [ln for ln in open(path).read().splitlines() if ln.strip().endswith("*")]
                if cur:
                    out[key] = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit()))
                    break
        except Exception:                                          # noqa: BLE001  (not readable on every box)
            pass
    return out


PMC_CLASS_KERNELS = None


def pmc_entry_matches_library(cls, entry):
    """a committed PMC entry is only attached to a bench row when every kernel it names is one the LOADED library runs for that class
    (lic360_codec_kernel_names): traffic of a kernel that no longer exists must not be reported as this run's"""
    global PMC_CLASS_KERNELS
    if PMC_CLASS_KERNELS is None:
        from lic360 import _lib
        PMC_CLASS_KERNELS = {}
        for part in _lib.lic360_codec_kernel_names().decode().split(";"):
            k, v = part.split("=")
            PMC_CLASS_KERNELS[k] = set(x.strip() for x in v.split("+"))
    have = PMC_CLASS_KERNELS.get(cls)
    names = set(x.strip() for x in str(entry.get("kernel", "")).split("+"))
    return bool(have) and bool(names) and names <= have


def newest_pmc_traffic():
    """(dict, file name) of the newest committed PMC collection profiles/rNN_pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes, tools/collect_profiles.sh), newest round first"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                return json.load(f), os.path.basename(path)
        except Exception:                                          # noqa: BLE001
            continue
    return {}, None


def dry_run(args):
    """gloo rehearsal of everything around the codec: rank set-up, sharding of config 4's image list, fences, MAX timing."""
    import lic360_shard as shard
    import torch.distributed as dist
    rank, local, world = shard.init_from_env("gloo")
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    mine = shard.shard_indices(64, rank, world)
    dt = shard.timed(lambda: time.sleep(0.01 * len(mine) / 8.0), args.steps)
    parts = shard.gather_results({i: rank for i in mine}, 64)
    if rank == 0:
        assert sorted(set(parts)) == list(range(world))
        print(json.dumps({"metric": "ERP Mpixels/s enc+dec @512x1024 model-idx 3; bitstream bit-exact vs ref", "value": 0.0, "unit": "Mpixel/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
                          "config": {"workload": "dry run: launch / sharding / reporting path only", "images_per_rank_config4": len(mine)}}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def run_rank(args):
    global MASKS
    MASKS = args.masks
    # numpy's / torch's host pools; the oracle's own pool is set per leg by cpu_baseline (orc_set_num_threads: all host cores, then 16)
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, host_cpu()[0])))
    import numpy as np
    import torch
    import torch.distributed as dist
    import lic360_shard as shard
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    rank, local, world = shard.init_from_env("nccl")          # one process per GPU; "nccl" is RCCL on ROCm
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d: start the ranks with `bench.py --gpus N` or torch.distributed.run" % (world, args.gpus)
    dev = torch.device("cuda", local)

    from util import make_main_params, make_imp_params          # seeded synthetic weights (numpy only)
    from lic360_fused import FusedCodec, FusedImpCodec
    wseed = 1000 * SSIM + MODEL_IDX
    layers, imp_layers = make_main_params(wseed, G), make_imp_params(wseed)
    B = args.batch
    ns = max(1, min(args.streams, B))
    sizes = [B // ns + (1 if i < B % ns else 0) for i in range(ns)]
    code_np, mask_np, level_np = synth_latents(B, seed0=1000 * rank)
    codecs, icodecs, codes, masks, levels, streams, istreams, mbufs, mdone = [], [], [], [], [], [], [], [], []
    o = 0
    for sz in sizes:
        c = FusedCodec(G, H, W, max_batch=sz, device=local)
        c.load_layers(layers)
        ic = FusedImpCodec(H // 2, W // 2, max_batch=sz, device=local)
        ic.load_layers(imp_layers)
        codecs.append(c)
        icodecs.append(ic)
        codes.append(torch.from_numpy(code_np[o:o + sz]).to(dev))
        masks.append(torch.from_numpy(mask_np[o:o + sz]).to(dev))
        levels.append(torch.from_numpy(level_np[o:o + sz]).to(dev))
        streams.append(torch.cuda.Stream(device=dev))
        # the importance-map codec of a sub-batch runs on a stream of its own: the two bitstreams of an image are independent,
        # and its small kernels fill the gaps the latent codec's launches leave
        istreams.append(torch.cuda.Stream(device=dev) if args.imp_streams else streams[-1])
        mbufs.append([torch.zeros((sz, G, H, W), dtype=torch.float32, device=dev) for _ in range(2)])   # the masks the importance decode derives on the device
        mdone.append([torch.cuda.Event(), torch.cuda.Event()])
        o += sz
    torch.cuda.synchronize(dev)

    flip, last_mb = [0], {}

    def run(cds, mks, lvs, imp=True):
        """encode then decode of one list of sub-batches (one per stream); inputs resident in HBM, bitstreams stay in HBM"""
        for c, ic, cd, mk, lv, st, ist in zip(codecs, icodecs, cds, mks, lvs, streams, istreams):
            if cd.shape[0]:
                if imp:
                    with torch.cuda.stream(ist):
                        ic.encode_async(lv)
                with torch.cuda.stream(st):
                    c.encode_async(cd, mk)
        flip[0] ^= 1
        for c, ic, cd, mk, lv, st, ist, mb2, ev2 in zip(codecs, icodecs, cds, mks, lvs, streams, istreams, mbufs, mdone):
            mb, ev = mb2[flip[0]], ev2[flip[0]]                             # two mask buffers per codec: the map decode of one step does not
            last_mb[id(c)] = mb                                             # wait for the latent decode of the step before
            n = cd.shape[0]
            if not n:
                continue
            if not imp:
                with torch.cuda.stream(st):
                    c.decode_async(mk, n)
                continue
            # the decoder's own data flow (test/lic360_demo.py:283-287, 218-238): the latent's mask is what the DECODED importance map
            # says.  The map decodes on its stream and writes the mask plane by plane; the latent decode runs behind it, each plane's
            # table kernel waiting for the event that covers its positions (lic360_codec_decode_gated).
            if ist is st:
                with torch.cuda.stream(st):
                    ic.decode_masked_async(n, mb[:n])                        # one stream: plain order, mask from the finished map
                    c.decode_async(mb[:n], n)
            else:
                ist.wait_event(ev)                                           # the previous latent decode may still be reading the mask buffer
                with torch.cuda.stream(ist):
                    gate = ic.decode_masked_async(n, mb[:n])
                with torch.cuda.stream(st):
                    c.decode_async(mb[:n], n, gate=gate)
                    ev.record(st)

    def exact(cds, mks, lvs, imp=True):
        ok = True
        for c, ic, cd, mk, lv in zip(codecs, icodecs, cds, mks, lvs):
            n = cd.shape[0]
            if not n:
                continue
            mb = last_mb.get(id(c))
            flags = {"code": bool(torch.equal(c.code_out[:n], cd * mk)), "err": int(c.err[:n].abs().sum().item()) == 0}
            if imp:                                                          # ... and the mask the device derived from the decoded map is the encoder's
                flags.update(levels=bool(torch.equal(ic.levels_out[:n], lv)), imp_err=int(ic.err[:n].abs().sum().item()) == 0,
                             mask=mb is not None and bool(torch.equal(mb[:n], mk)))
            if not all(flags.values()):
                bad = (c.code_out[:n] != cd * mk).flatten(1).any(1).nonzero().flatten().tolist()
                sys.stderr.write("[bench] round trip NOT exact on rank %d, sub-batch of %d images: %s; images with wrong symbols: %s\n" % (rank, n, flags, bad[:16]))
                # which direction: the bitstreams of the failed pass against a fresh encode of the same sub-batch alone on the GPU, then a decode alone
                torch.cuda.synchronize(dev)
                nb0, by0 = c.nbytes[:n].clone(), c.bytes[:n].clone()
                c.encode_async(cd, mk)
                torch.cuda.synchronize(dev)
                same = [bool(nb0[i] == c.nbytes[i]) and bool(torch.equal(by0[i, :int(nb0[i])], c.bytes[i, :int(nb0[i])])) for i in range(n)]
                c.decode_async(mk, n)
                torch.cuda.synchronize(dev)
                again = (c.code_out[:n] != cd * mk).flatten(1).any(1).nonzero().flatten().tolist()
                sys.stderr.write("[bench]   images whose bitstream differs from a fresh encode: %s; wrong after a decode alone: %s\n" % ([i for i in range(n) if not same[i]][:16], again[:16]))
            ok = ok and all(flags.values())
        return ok

    say = progress if rank == 0 else (lambda m: None)
    calib = calibrate(dev) if rank == 0 else {}
    say("codecs ready; calibration %s; warm-up + %d timed steps of %d images" % (calib, args.steps, B))
    step = lambda: run(codes, masks, levels, True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    ok = exact(codes, masks, levels)
    dt = shard.timed(step, args.steps, dev)                   # barrier + sync both sides, MAX over ranks; instrumentation off
    ok = ok and exact(codes, masks, levels)
    nbytes = np.concatenate([c.nbytes[:cd.shape[0]].cpu().numpy() for c, cd in zip(codecs, codes)])
    inbytes = np.concatenate([c.nbytes[:cd.shape[0]].cpu().numpy() for c, cd in zip(icodecs, codes)])
    img0 = (bytes(codecs[0].bytes[0, :int(nbytes[0])].cpu().numpy().tobytes()), bytes(icodecs[0].bytes[0, :int(inbytes[0])].cpu().numpy().tobytes()))

    say("timed region done: %.1f ms per step" % (dt / args.steps * 1e3))
    extras = {}
    ks = max(1, min(args.steps, 3))                                    # timed steps of the side figures (the headline alone runs args.steps: the default run stays within minutes)
    if not args.no_extras and B > 144 and all(sz >= 48 for sz in sizes) and ns == 3:
        # the per-GPU batch of rounds 1-5 (144 = 3 x 48 images) through the same codecs: what the larger sub-batches add
        c48, m48, l48 = [c[:48] for c in codes], [m[:48] for m in masks], [l[:48] for l in levels]
        run(c48, m48, l48)
        dt48 = shard.timed(lambda: run(c48, m48, l48), ks, dev)
        ok = ok and exact(c48, m48, l48)
        extras["batch144"] = {"value": world * 144 * ks * PIXELS / dt48 / 1e6, "unit": "Mpixel/s", "ms_per_step": dt48 / ks * 1e3,
                              "note": "144 images per GPU (3 sub-batches of 48), the per-GPU batch of rounds 1-5"}
    if not args.no_extras and args.masks != "iid":
        # the same step on the masks of rounds 1-5 (every map cell drawn independently): nothing for the dead-cone skip to find -- the adversarial case
        ci, mi, li = synth_latents(B, seed0=1000 * rank, kind="iid")
        o, cdi, mki, lvi = 0, [], [], []
        for sz in sizes:
            cdi.append(torch.from_numpy(ci[o:o + sz]).to(dev)); mki.append(torch.from_numpy(mi[o:o + sz]).to(dev)); lvi.append(torch.from_numpy(li[o:o + sz]).to(dev))
            o += sz
        run(cdi, mki, lvi)
        dti = shard.timed(lambda: run(cdi, mki, lvi), ks, dev)
        ok = ok and exact(cdi, mki, lvi)
        extras["iid_masks"] = {"value": world * B * ks * PIXELS / dti / 1e6, "unit": "Mpixel/s", "ms_per_step": dti / ks * 1e3,
                               "note": "the timed step on i.i.d. importance maps (tests/util.py:latent, the workload of rounds 1-5)"}
        del cdi, mki, lvi
    if not args.no_extras:
        # the latent stream alone (98.6 % of the bytes, 95 % of the MACs): round 1's headline, kept for comparison
        run(codes, masks, levels, False)
        dt_lat = shard.timed(lambda: run(codes, masks, levels, False), ks, dev)
        extras["latent_stream_only"] = {"value": world * B * ks * PIXELS / dt_lat / 1e6, "unit": "Mpixel/s", "ms_per_step": dt_lat / ks * 1e3}
        # BASELINE.json configs[3]: 64 images, image i -> rank i mod N, through the same codecs and streams
        mine = shard.shard_indices(64, rank, world)
        c4, m4, l4 = synth_latents(64, seed0=640000)
        sz4 = split_for_streams(len(mine), ns)
        per = [mine[sum(sz4[:i]):sum(sz4[:i + 1])] for i in range(ns)]
        cd4 = [torch.from_numpy(c4[ix]).to(dev) if ix else torch.zeros((0, G, H, W), device=dev) for ix in per]
        mk4 = [torch.from_numpy(m4[ix]).to(dev) if ix else torch.zeros((0, G, H, W), device=dev) for ix in per]
        lv4 = [torch.from_numpy(l4[ix]).to(dev) if ix else torch.zeros((0, 1, H // 2, W // 2), device=dev) for ix in per]
        run(cd4, mk4, lv4)
        dt4 = shard.timed(lambda: run(cd4, mk4, lv4), 3, dev)
        ok = ok and exact(cd4, mk4, lv4)
        extras["config4"] = {"images": 64, "images_per_gpu": len(mine), "ms": dt4 / 3 * 1e3, "value": 64 * PIXELS / (dt4 / 3) / 1e6, "unit": "Mpixel/s",
                             "scaling": "strong", "note": "BASELINE.json configs[3]: fixed list of 64 images, image i -> rank i mod N, both streams"}
        if world == 1:
            # what ONE rank of BASELINE.json configs[3] does at 8 GPUs: its 8 images of the 64 (both streams, same codecs and HIP
            # streams).  That regime is latency-bound (238 dependent planes, one serial coder chain per image), so 64 images over
            # eight such ranks is the honest prediction of the 8-GPU strong-scaling figure -- the driver measures the real one.
            sz8 = split_for_streams(8, ns)
            ix8 = [list(range(sum(sz8[:i]), sum(sz8[:i + 1]))) for i in range(ns)]
            cd8 = [torch.from_numpy(c4[ix]).to(dev) if ix else torch.zeros((0, G, H, W), device=dev) for ix in ix8]
            mk8 = [torch.from_numpy(m4[ix]).to(dev) if ix else torch.zeros((0, G, H, W), device=dev) for ix in ix8]
            lv8 = [torch.from_numpy(l4[ix]).to(dev) if ix else torch.zeros((0, 1, H // 2, W // 2), device=dev) for ix in ix8]
            run(cd8, mk8, lv8)
            dt8 = shard.timed(lambda: run(cd8, mk8, lv8), 3, dev)
            ok = ok and exact(cd8, mk8, lv8)
            extras["config4_per_gpu_share"] = {"images": 8, "ms": dt8 / 3 * 1e3, "value": 8 * PIXELS / (dt8 / 3) / 1e6, "unit": "Mpixel/s",
                                               "predicted_8gpu_strong": {"value": 64 * PIXELS / (dt8 / 3) / 1e6, "unit": "Mpixel/s",
                                                                         "vs_this_gpu_on_all_64": (64 * PIXELS / (dt8 / 3)) / (64 * PIXELS / (dt4 / 3))},
                                               "note": "8 images alone on one GPU = the per-rank share of configs[3] at N = 8; latency-bound"}
        # BASELINE.json configs[4]: LIC3602K 1024x2048 ERPs (48x128x256 latents, 32x64 -> 64x128 importance maps), model-idx 7 seed,
        # 16 images per stream through codecs of their own (decode order runs on 64-row windows there, DESIGN.md 4.1 b)
        # (set-up may fail on one rank only, e.g. out of memory: the collective steps below run on every rank or on none)
        say("config 4 figures done; config 5 (1024x2048 ERPs)")
        H5, W5, B5 = 2 * H, 2 * W, 16
        c5 = i5 = cd5 = mk5 = lv5 = None
        err5 = None
        try:
            l5, il5 = make_main_params(1000 * SSIM + 7, G), make_imp_params(1000 * SSIM + 7)
            c5 = [FusedCodec(G, H5, W5, max_batch=B5, device=local) for _ in range(ns)]
            i5 = [FusedImpCodec(H5 // 2, W5 // 2, max_batch=B5, device=local) for _ in range(ns)]
            for c in c5:
                c.load_layers(l5)
            for c in i5:
                c.load_layers(il5)
            cd, mk, lv = synth_latents(B5 * ns, seed0=5000000 + 1000 * rank, h=H5, w=W5)
            cd5 = [torch.from_numpy(cd[i * B5:(i + 1) * B5]).to(dev) for i in range(ns)]
            mk5 = [torch.from_numpy(mk[i * B5:(i + 1) * B5]).to(dev) for i in range(ns)]
            lv5 = [torch.from_numpy(lv[i * B5:(i + 1) * B5]).to(dev) for i in range(ns)]
            torch.cuda.synchronize(dev)
        except Exception as e:                                         # noqa: BLE001  (never lose the headline to a side figure)
            err5 = repr(e)[:300]
        if shard.all_ok(err5 is None, dev):
            mb5 = [torch.zeros_like(m) for m in mk5]                         # decode side: the mask comes from the decoded map (see run())
            ev5 = [torch.cuda.Event() for _ in mk5]
            def run5():
                for c, ic, a, b_, l_, st, ist in zip(c5, i5, cd5, mk5, lv5, streams, istreams):
                    with torch.cuda.stream(ist):
                        ic.encode_async(l_)
                    with torch.cuda.stream(st):
                        c.encode_async(a, b_)
                for c, ic, st, ist, mb, ev in zip(c5, i5, streams, istreams, mb5, ev5):
                    if ist is st:
                        with torch.cuda.stream(st):
                            ic.decode_masked_async(B5, mb)
                            c.decode_async(mb, B5)
                    else:
                        ist.wait_event(ev)
                        with torch.cuda.stream(ist):
                            gate = ic.decode_masked_async(B5, mb)
                        with torch.cuda.stream(st):
                            c.decode_async(mb, B5, gate=gate)
                            ev.record(st)
            run5()
            dt5 = shard.timed(run5, 2, dev)
            ok5 = all(bool(torch.equal(c.code_out[:B5], a * b_)) and int(c.err[:B5].abs().sum().item()) == 0 for c, a, b_ in zip(c5, cd5, mk5))
            ok5 = ok5 and all(bool(torch.equal(mb, m)) for mb, m in zip(mb5, mk5))
            ok5 = ok5 and all(bool(torch.equal(ic.levels_out[:B5], l_)) and int(ic.err[:B5].abs().sum().item()) == 0 for ic, l_ in zip(i5, lv5))
            ok = ok and ok5
            extras["config5"] = {"images_per_gpu": B5 * ns, "ms": dt5 / 2 * 1e3, "value": world * B5 * ns * 4 * PIXELS / (dt5 / 2) / 1e6, "unit": "Mpixel/s",
                                 "roundtrip_exact": ok5, "mean_latent_bytes": float(np.mean([float(c.nbytes[:B5].float().mean().item()) for c in c5])),
                                 "note": "BASELINE.json configs[4]: 1024x2048 ERPs (48x128x256 latents), model-idx 7 seed, both streams, %d images on %d streams" % (B5 * ns, ns)}
        else:
            extras["config5"] = {"error": err5 or "set-up failed on another rank"}
        del c5, i5, cd5, mk5, lv5
        mb5 = None
        torch.cuda.empty_cache()
    ok = shard.all_ok(ok, dev)

    if rank == 0:
        out = {
            "metric": "ERP Mpixels/s enc+dec @512x1024 model-idx 3; bitstream bit-exact vs ref",
            "value": world * B * args.steps * PIXELS / dt / 1e6, "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d/GPU synthetic 512x1024 ERPs, model-idx 3 --ssim seed, latent+importance bitstreams, entropy enc+dec" % B,
                       "images_per_gpu_per_step": B, "streams": ns, "roundtrip_exact": ok,
                       "bit_exact_against": "CPU oracle (arithmetic coder pinned to the reference's ArithmeticCoder.cpp; float kernels unpinnable without CUDA)",
                       "mean_latent_bytes": float(nbytes.mean()), "mean_importance_bytes": float(inbytes.mean())},
        }
        assert len(out["config"]["workload"]) <= 120
        # the side figures as SCALAR keys of `config` (the driver's record keeps scalars only); the nested forms stay beside them
        flat = out["config"]
        flat.update(calib)
        if calib.get("calib_mfma_tflops"):
            flat["value_per_calib_mfma_tflop"] = out["value"] / calib["calib_mfma_tflops"]
        if "iid_masks" in extras:
            flat["iid_masks_mpixel_s"] = extras["iid_masks"]["value"]
        if "batch144" in extras:
            flat["batch144_mpixel_s"] = extras["batch144"]["value"]
        if "latent_stream_only" in extras:
            flat["latent_only_mpixel_s"] = extras["latent_stream_only"]["value"]
        if "config4" in extras:
            flat["config4_mpixel_s"], flat["config4_ms"] = extras["config4"]["value"], extras["config4"]["ms"]
        if "config4_per_gpu_share" in extras:
            sh = extras["config4_per_gpu_share"]
            flat["config4_share8_ms"] = sh["ms"]
            flat["config4_pred_8gpu_mpixel_s"] = sh["predicted_8gpu_strong"]["value"]
            flat["config4_pred_8gpu_x"] = sh["predicted_8gpu_strong"]["vs_this_gpu_on_all_64"]
        if "value" in extras.get("config5", {}):
            flat["config5_mpixel_s"], flat["config5_ms"] = extras["config5"]["value"], extras["config5"]["ms"]
        out["config"].update(extras)
        say("instrumented passes (per-kernel-class events)")
        inst = instrumented(args, codecs, icodecs, codes, masks, levels, streams, dev, dt, B)
        skip = inst.pop("skip")
        out.update(inst)
        out["config"]["masks"] = args.masks
        out["config"]["dead_mac_fraction"] = skip["dead_mac_fraction"]
        out["config"]["dead_cone_skip"] = skip
        if world == 1 and not args.no_extras:
            si = single_image(codecs[0], icodecs[0], codes[0], masks[0], levels[0], streams[0], istreams[0], mbufs[0][0], dev)
            out["config"]["single_image"] = si
            out["config"]["single_encode_ms"], out["config"]["single_decode_ms"] = si["encode_ms"], si["decode_ms"]
            say("single image done; streaming ops")
            try:
                import stream_ops_bench
                rows = stream_ops_bench.measure(batches=(32,), device=local)
                out["kernels"] += [{"kernel": r["op"], "bound": "hbm", "achieved": r["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": r["frac_of_hbm_peak"], "avg_launch_ms": r["us"] * 1e-3, "images_per_launch": 32} for r in rows]
            except Exception as e:                                     # noqa: BLE001  (never lose the headline to a side table)
                out["config"]["streaming_ops_error"] = repr(e)
            try:       # the neighbouring row f1: analysis / synthesis transforms (library convs + native ops) and the one-pass GDN
                say("transforms + whole codec")
                import transform_bench
                out["kernels"] += transform_bench.measure(batch=8, device=local)
            except Exception as e:                                     # noqa: BLE001
                out["config"]["transform_bench_error"] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(layers, imp_layers, code_np[0:1], mask_np[0:1], level_np[0:1], img0[0], img0[1])
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def single_image(c, ic, code, mask, lv, st, ist, mb, dev):
    """BASELINE.json configs[1] / [2]: one 512x1024 image alone on the GPU -- encode latency, decode latency (both streams).
    Encode: the two bitstreams are independent (two HIP streams).  Decode: the latent's mask is derived on the device from the decoded
    importance map; the latent decode runs behind the map's decode, gated plane by plane (lic360_codec_decode_gated)."""
    import torch
    res = {}
    ev = torch.cuda.Event()

    def enc():
        ist.wait_stream(st)
        with torch.cuda.stream(ist):
            ic.encode_async(lv[:1])
        with torch.cuda.stream(st):
            c.encode_async(code[:1], mask[:1])
            st.wait_stream(ist)

    def dec():
        if ist is st:
            with torch.cuda.stream(st):
                ic.decode_masked_async(1, mb[:1])
                c.decode_async(mb[:1], 1)
            return
        ist.wait_stream(st)                                                  # one image after the other: this measures a latency
        with torch.cuda.stream(ist):
            gate = ic.decode_masked_async(1, mb[:1])
        with torch.cuda.stream(st):
            c.decode_async(mb[:1], 1, gate=gate)
            st.wait_stream(ist)

    for name, fn in (("encode_ms", enc), ("decode_ms", dec)):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        res[name] = (time.perf_counter() - t0) / 3 * 1e3
    res["roundtrip_exact"] = bool(torch.equal(c.code_out[:1], code[:1] * mask[:1]) and torch.equal(ic.levels_out[:1], lv[:1]) and torch.equal(mb[:1], mask[:1]))
    res["value"] = PIXELS / ((res["encode_ms"] + res["decode_ms"]) * 1e-3) / 1e6
    res["unit"] = "Mpixel/s"
    res["note"] = "configs[1]/[2]: latency-bound (238 dependent planes x 14 launches, one serial coder chain); decode: mask from the decoded importance map, latent decode gated behind the map's"
    return res


def chain_macs(layer):
    """MACs per output POSITION of output group g of one net's layer (mask rule, extension/cconv_ec_cuda.cu:288-290; image borders ignored: the
    figures below are only ever used as ratios)"""
    import numpy as np
    hidden, cin, cout = (0, 1, 4) if layer == 0 else ((1, 4, 3) if layer == 11 else (1, 4, 4))
    g = np.arange(G)[:, None, None]
    kh, kw = np.arange(5)[None, :, None], np.arange(5)[None, None, :]
    return np.clip(g + 4 - kh - kw + hidden, 0, G).sum((1, 2)).astype(np.float64) * cin * cout


def executed_fractions(stat_pairs, images):
    """{kernel class: executed / algorithmic MACs} of the latent nets from the codecs' skip counters (lic360_codec_skip_stats; csrc/need.h): what the
    dead-cone skip left of the reference's work.  stat_pairs: [(enc [12, 64], dec [12, 64], active)] per codec, images: images per codec."""
    import numpy as np
    ntiles = ((H + 3) // 4) * ((W + 15) // 16)
    num = {k: 0.0 for k in ("ec_first", "ec_hidden", "ec_last", "dc_first", "dc_hidden", "dc_last")}
    den = dict(num)
    for (enc, dec, active), b in zip(stat_pairs, images):
        if not b:
            continue
        for l in range(12):
            cm = chain_macs(l)
            full = 3.0 * b * H * W * cm.sum()
            cls = "first" if l == 0 else ("last" if l == 11 else "hidden")
            for side in ("ec_", "dc_"):
                den[side + cls] += full
            # encode order: live (tile, group block) pairs, 64 positions each, of N samples (last layer: the three nets of a pair run as phases)
            if active >= 1 and l > 0:
                gpb, rep = (5, 3.0) if l == 11 else (4, 1.0)
                wgb = np.array([cm[i:i + gpb].sum() for i in range(0, G, gpb)])
                num["ec_" + cls] += rep * 64.0 * float((enc[l, :len(wgb)].astype(np.float64) * wgb).sum()) * (H * W / (64.0 * ntiles))
            else:
                num["ec_" + cls] += full
            if active >= 2 and l > 0 and b % 8 == 0 and b >= 16:
                num["dc_" + cls] += float((dec[l, :G].astype(np.float64) * cm).sum())
            else:
                num["dc_" + cls] += full
    return {k: (num[k] / den[k] if den[k] else 1.0) for k in num}


def instrumented(args, codecs, icodecs, codes, masks, levels, streams, dev, dt, B):
    """Per-kernel-class durations from HIP events on the launch stream: (a) one sub-batch ALONE on the GPU, (b) all streams."""
    import torch
    b0 = int(codes[0].shape[0])

    def one_pass(k):
        for i in k:
            with torch.cuda.stream(streams[i]):
                codecs[i].encode_async(codes[i], masks[i])
        for i in k:
            with torch.cuda.stream(streams[i]):
                codecs[i].decode_async(masks[i], codes[i].shape[0])
        torch.cuda.synchronize(dev)

    # what the dead-cone skip executes (an untimed pass of its own: the counters are atomics): sub-batch 0 for the isolated rows, all for the step
    for c in codecs:
        c.skip_stats(True)
    one_pass(range(len(codecs)))
    stat_pairs = [c.skip_stats(False, read=True) + (c.skip_active(),) for c in codecs]
    imgs = [int(cd.shape[0]) for cd in codes]
    ex0 = executed_fractions(stat_pairs[:1], imgs[:1])
    ex_all = executed_fractions(stat_pairs, imgs)
    codecs[0].profile(True)
    one_pass([0])
    iso = codecs[0].profile_read()
    codecs[0].profile(False)
    for c in codecs:
        c.profile(True)
    one_pass(range(len(codecs)))
    conc = {}
    for c in codecs:
        for k, (ms, n) in c.profile_read().items():
            a = conc.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
        c.profile(False)

    rows = []
    iso_live = {k for k, (ms, n) in iso.items() if n}
    for cls, (ms, n) in iso.items():
        if n == 0:
            continue
        row = {"kernel": cls, "launches": n, "avg_launch_ms": ms / n, "images_per_launch": b0, "total_ms": ms}
        part = cls.split("_", 1)[1] if cls[:3] in ("ec_", "dc_") else None
        if part in GMAC:
            fl = 2 * GMAC[part] * 1e9 * b0                         # whole pass of this class over b0 images, one direction: the reference's count
            fx = fl * ex0.get(cls, 1.0)                            # ... and what was executed of it (dead-cone skip): achieved / frac are priced on THIS
            row.update(bound="mfma", achieved=fx / (ms * 1e-3) / 1e12, peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                       algorithmic_flops_per_launch=fl / n, executed_flops_per_launch=fx / n, executed_fraction=ex0.get(cls, 1.0),
                       reference_work_rate={"value": fl / (ms * 1e-3) / 1e12, "unit": "TFLOP/s", "note": "the reference's MACs of this pass / its time (skipped work counted as done: a speed, not an efficiency)"})
        elif cls in ("enc_tables", "dec_tables"):
            per_sym = 52.0 if cls == "enc_tables" else 56.0        # 9 floats of net output + symbol/mask in; (lo,hi) record or 7 packed 16-bit entries + flag out
            by = per_sym * NSYM * b0
            row.update(bound="hbm", achieved=by / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", algorithmic_bytes_per_launch=by / n)
        else:                                                      # serial arithmetic-coder chains: one wave per image
            row.update(bound="latency", achieved=ms * 1e6 / NSYM, peak=None, unit="ns per symbol of one image's serial chain")
        row["frac"] = row["achieved"] / row["peak"] if row.get("peak") else None
        if cls in conc and conc[cls][1]:
            row["avg_launch_ms_concurrent"] = conc[cls][0] / conc[cls][1]
        rows.append(row)
    convs = [r for r in rows if r["bound"] == "mfma"]
    dom = max(convs, key=lambda r: r["total_ms"])
    # the importance-map stream of the same sub-batch, alone on the GPU: whole passes (12 conv layers of the one-group 144-channel net -- layer 0
    # on the generic kernel, 1..11 on k_cconv144 --, 49-way tables, coder; decode: 95 planes x 13 launches), wall time between synchronisations
    ic0, lv0 = icodecs[0], levels[0]
    for name, fn in (("imp_ec", lambda: ic0.encode_async(lv0)), ("imp_dc", lambda: ic0.decode_async(b0))):
        with torch.cuda.stream(streams[0]):
            fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        fl = 2 * IMP_GMAC * 1e9 * b0
        rows.append({"kernel": name, "launches": 1, "avg_launch_ms": ms, "images_per_launch": b0, "total_ms": ms, "bound": "mfma",
                     "achieved": fl / (ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                     "algorithmic_flops_per_launch": fl,
                     "note": "whole pass of the importance-map %s of %d maps (all conv layers + tables + coder), wall time; %s" % (
                         "encode" if name == "imp_ec" else "decode", b0,
                         "one launch per layer" if name == "imp_ec" else "95 planes x 13 dependent launches: launch-latency bound")})
    # fabric-side bytes per launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, tools/collect_profiles.sh;
    # Infinity-Cache hits are counted), scaled to this run's images per launch
    pmc, pmc_file = newest_pmc_traffic()
    for r in rows:
        pm = pmc.get(r["kernel"])
        if isinstance(pm, dict):
            if pmc_entry_matches_library(r["kernel"], pm):
                r["traffic"] = pm["traffic_bytes_per_launch"] * b0 / pm["images_per_launch"]
                if pmc.get("masks") and pmc.get("masks") != args.masks:
                    r["traffic_note"] = "collected on %s masks" % pmc.get("masks")
            else:
                r["traffic"] = None
                r["traffic_note"] = "profiles/%s names kernels (%s) the loaded library does not run for this class: re-collect (tools/collect_profiles.sh)" % (pmc_file, pm.get("kernel"))
        if r["kernel"] == "ec_last" and "enc_tables" not in iso_live:
            # the fused last layer + CDF-table kernel (N1): HBM view next to the MFMA view.  Algorithmic bytes: 3 nets x 4 channels in,
            # symbol + mask in, one (cdf[sym], cdf[sym+1]) record out = 64 B per symbol, + the layer's weights once per launch
            # (SURVEY.md 8d prices the unfused form, which writes the whole 9-entry row, at 113 B per symbol)
            by = 64.0 * NSYM * b0 + 8.29e6
            r["fused_with"] = "softmax / sigma floor / erf CDF / fix-up / (cdf[sym], cdf[sym+1]) record write (no y round trip, no enc_tables launch)"
            r["hbm_view"] = {"algorithmic_bytes_per_launch": by, "bytes_per_symbol": by / (NSYM * b0), "achieved": by / (r["avg_launch_ms"] * 1e-3) / 1e9,
                             "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": by / (r["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    traffic = dom.get("traffic")
    conv_ms = sum(r["total_ms"] for r in convs)
    cls_gmac = {"ec_first": GMAC["first"], "ec_hidden": GMAC["hidden"], "ec_last": GMAC["last"], "dc_first": GMAC["first"], "dc_hidden": GMAC["hidden"], "dc_last": GMAC["last"]}
    exec_net0 = sum(cls_gmac[k] * ex0[k] for k in cls_gmac)            # executed GMAC per image, both directions (sub-batch 0 / whole batch)
    exec_net = sum(cls_gmac[k] * ex_all[k] for k in cls_gmac)
    roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": dom["frac"], "traffic": traffic,
                "traffic_source": ("profiles/%s (committed rocprofv3 --pmc passes of tools/collect_profiles.sh, not collected in this run), scaled to "
                                   "%d images per launch" % (pmc_file, b0)) if traffic is not None else None,
                "algorithmic_flops_per_launch": dom["algorithmic_flops_per_launch"], "executed_flops_per_launch": dom["executed_flops_per_launch"],
                "executed_fraction": dom["executed_fraction"],
                "accounting": "achieved / frac = EXECUTED flops (the reference's count minus what the dead-cone skip did not compute, from the codec's own counters) / kernel time",
                "images_per_launch": b0, "avg_launch_ms": dom["avg_launch_ms"], "launches": dom["launches"],
                "how": "one sub-batch alone on the GPU, one stream, HIP events around every launch on the launch stream (instrumented pass, "
                       "outside the timed region)",
                "rocprof_kernel_names": "dc_hidden / dc_last launches run as k_cconv4v6l<4> (task records from the dead-cone lists: batches of >= 16 images) -- "
                                        "or, without lists, k_cconv4v6t<4> (planes that tape-pack their samples) / k_cconv4v6<4, false, false> (full-length planes)",
                "concurrent": {"avg_launch_ms": dom.get("avg_launch_ms_concurrent"), "streams": len(codecs),
                               "note": "same kernel while the other streams' kernels share the CUs (the timed region's regime)"},
                "all_conv_layers_isolated": {"achieved": 2 * exec_net0 * 1e9 * b0 / (conv_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                                             "frac": 2 * exec_net0 * 1e9 * b0 / (conv_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS},
                "whole_step": {"achieved": 2 * (exec_net + 2 * IMP_GMAC) * 1e9 * B / (dt / args.steps) / 1e12, "unit": "TFLOP/s",
                               "frac": 2 * (exec_net + 2 * IMP_GMAC) * 1e9 * B / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                               "note": "EXECUTED FLOPs of every conv layer (both nets, both directions) / wall time of the timed step"}}
    skip = {"dead_mac_fraction": 1.0 - exec_net / (2 * NET_GMAC), "executed_fraction_by_class": ex_all,
            "skip_active": [p[2] for p in stat_pairs],
            "note": "share of the latent nets' MACs (reference count, both directions, whole per-GPU batch) that no coded symbol can observe AND that the "
                    "kernels' granularity (encode: 4 x 16 tile x group block; decode: hull of a sample's live rows per three-group block and plane) let "
                    "them skip; csrc/need.h"}
    return {"roofline": roofline, "kernels": rows, "skip": skip}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # supervisor: HIP-free by construction -- it imports neither torch nor the HIP library; devices are counted from the
        # KFD topology in sysfs (a rank whose device is missing fails in set_device and the launcher exits non-zero)
        import lic360_shard as shard
        if not args.dry_run:
            have = shard.count_gpus_sysfs()
            if have is not None and have < args.gpus:
                sys.stderr.write("bench.py: --gpus %d but only %d GPU nodes in /sys/class/kfd\n" % (args.gpus, have))
                return 2
        return shard.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)
    return dry_run(args) if args.dry_run else run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
