"""GPU box: randomized soak of the decode-order kernel's sample packing (tape packing, round 5) -- lic360_cconv4_dc_plane plane by plane against the
oracle (tests/test_gpu_ops.py: _cconv4_dc_planes) on random layer shapes: groups, image size up to 64 rows, samples = 8 * tape * k per net so that the
tape engages (tapes of 2..6 samples), 1..3 nets, hidden / first / last layers.  usage: dc_tape_soak.py [cases] [seed]; prints one line per case."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lic360
import test_gpu_ops as T

ncases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
for i in range(ncases):
    G = int(rng.integers(3, 25))
    H = int(rng.integers(5, 65))
    W = int(rng.integers(5, 40))
    nb = int(rng.integers(1, 4))
    per_net = 8 * int(rng.choice([2, 3, 4, 5, 6, 9, 10, 12]))
    layer = int(rng.integers(0, 3))                                    # 0 hidden (4 -> 4), 1 last (4 -> 3, no PReLU), 2 first (1 -> 4)
    cout, hidden, act, cin = [(4, True, True, 4), (3, True, False, 4), (4, False, True, 1)][layer]
    while per_net * nb * G * cin * (H + W) * (H + 4) > 6e8:           # keep a case within a few seconds of oracle time
        per_net = max(16, per_net // 2)
        if per_net == 16:
            break
    case = (G, cout, hidden, act, nb, per_net * nb, H, W, cin)
    t0 = time.time()
    try:
        T._cconv4_dc_planes(lic360, case, "lic360_cconv4_dc_plane", ("lic360_conv4_supported", "lic360_conv4_packed_floats", "lic360_conv4_pack"))
        print("case %2d %s OK  %.1f s" % (i, case, time.time() - t0), flush=True)
    except AssertionError as e:
        bad += 1
        print("case %2d %s FAILED: %s" % (i, case, str(e)[:200]), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
