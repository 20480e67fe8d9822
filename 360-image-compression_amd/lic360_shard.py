"""Per-image data parallelism over the GPUs of one node: ERP images are independent, so image i
goes to rank i mod world, every rank owns one GPU and one codec, and NO data-path collective is needed.  The only
cross-rank operations are the barrier + MAX of the wall time used for throughput reporting and an optional gather
of per-image results (bitstream sizes / digests) to rank 0.  Works with backend "nccl" (= RCCL on ROCm) on GPUs and
"gloo" on CPU (tests)."""
import os
import socket
import subprocess
import sys
import time


class _Lazy:
    """torch / torch.distributed are imported on first use: the `bench.py --gpus N` supervisor imports this module for
    launch_ranks() and count_gpus_sysfs() only and must stay free of torch (hence of any HIP call)."""
    def __init__(self, name):
        self._name, self._mod = name, None

    def __getattr__(self, attr):
        if self._mod is None:
            import importlib
            self._mod = importlib.import_module(self._name)
        return getattr(self._mod, attr)


torch = _Lazy("torch")
dist = _Lazy("torch.distributed")


def count_gpus_sysfs(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPU agents of this machine from the KFD topology (nodes with simd_count > 0), read from sysfs: no HIP / HSA call,
    no /dev/kfd open.  None when the topology is not there (no amdgpu driver: nothing to check against)."""
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(root, d, "properties")) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
        except (OSError, ValueError):
            continue
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    if vis is not None and vis.strip() != "":
        n = min(n, len([t for t in vis.split(",") if t.strip() != ""]))
    return n


def shard_indices(n_items, rank, world):
    """Images handled by `rank`: i = rank, rank+world, ... (round-robin keeps shards within one image of each other)."""
    return list(range(rank, n_items, world))


def launch_ranks(script, argv, nproc, extra_env=None):
    """Start `nproc` ranks of `script argv...` on this node, one process per GPU, exactly as the driver does it:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script ...`.
    Call it from a process that has made NO HIP call yet (a parent that has touched the GPU must not be replaced, and a
    child forked after HIP initialisation inherits a broken context): the caller stays a plain supervisor and returns
    the launcher's exit code.  Rank 0's stdout is the caller's stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=env)


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by torch.distributed.run; returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def fence(device=None):
    """barrier + device synchronize on both sides of a timed region."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)


def timed(fn, steps, device=None):
    """Run fn() `steps` times between fences; returns the MAX wall time over ranks (seconds)."""
    fence(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence(device)
    dt = time.perf_counter() - t0
    return reduce_max(dt, device)


def reduce_max(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ok(flag, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return bool(flag)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def gather_results(local_results, n_items):
    """local_results: {image index: picklable}.  Returns the full list on rank 0 (None elsewhere)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [local_results[i] for i in range(n_items)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_results)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(n_items)]
