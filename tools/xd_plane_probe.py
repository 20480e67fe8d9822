"""lic360_cconv16_dc_plane alone: one hidden layer, N samples (default 144 = 48 images x 3 nets), chosen planes; per-launch time
from HIP events and -- in a diagnostic build with -DXD_STAMP -- the cycles each wave spent per phase."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import lic360 as lic
from util import conv_params
G, H, W, N = 48, 64, 128, int(os.environ.get("XN", 144))
planes = [int(v) for v in os.environ.get("XP", "30,70,110,150,190,220").split(",")]
L = lic._lib
rows, pitch, row0, col0 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
assert L.lic360_dc4_layout(H, W, C.byref(rows), C.byref(pitch), C.byref(row0), C.byref(col0)) == 0
SK = rows.value * pitch.value
plan = C.c_void_p(0)
assert L.lic360_conv_plan_create(G * 4, G, G * 4, 5, 6, C.byref(plan)) == 0
rng = np.random.default_rng(0)
w, b, a = conv_params(rng, 3, G * 4, G * 4, act=True)
wd, bd, ad = (torch.from_numpy(t).cuda() for t in (w, b, a))
packed = torch.empty(3 * L.lic360_conv16dc_packed_floats(plan), dtype=torch.float32, device="cuda:0")
s = lic._stream(0); P = lic._p
assert L.lic360_conv16dc_pack(s, plan, P(wd), 3, P(packed)) == 0
nfl = L.lic360_conv4_buffer_floats(0, N * G * 4, H, W)
NX = int(os.environ.get("XCOLD", 1))                    # > 1: rotate over this many input buffers, so that every launch reads cold activations
xs = [torch.randn(nfl, dtype=torch.float32, device="cuda:0") for _ in range(NX)]
x = xs[0]; out = torch.zeros(nfl, dtype=torch.float32, device="cuda:0")
packed4 = torch.empty(3 * L.lic360_conv4_packed_floats(plan), dtype=torch.float32, device="cuda:0")
assert L.lic360_conv4_pack(s, plan, P(wd), 3, P(packed4)) == 0
has_stamps = hasattr(L, "lic360_xd_stamps")
try:
    L.lic360_xd_stamps
except AttributeError:
    has_stamps = False
def run(fn, pk, p, reps=20):
    for i in range(3): assert fn(s, plan, P(xs[i % NX]), P(pk), P(bd), P(ad), None, P(out), N, H, W, 3, p, N) == 0, L.lic360_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(s, plan, P(xs[i % NX]), P(pk), P(bd), P(ad), None, P(out), N, H, W, 3, p, N)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
tot16 = tot4 = 0.0
for p in planes:
    if has_stamps:
        L.lic360_xd_stamps.argtypes = [C.c_void_p, C.c_int]; L.lic360_xd_stamps(None, 1)
    t16 = run(getattr(L, os.environ.get("XENTRY", "lic360_cconv16_dc_plane")), packed, p)
    line = "plane %3d: new %7.1f us" % (p, t16)
    if has_stamps:
        buf = (C.c_ulonglong * (256 * 8 * 10))()
        L.lic360_xd_stamps(buf, 0)
        st = np.array(buf, dtype=np.float64).reshape(256, 8, 10) / 23.0          # per launch (3 warm-up + 20 timed)
        tot = st.sum(-1)
        line += "  | wave cycles/launch: total %6.0f (max wg %6.0f)  setup %5.0f  kloop %6.0f  eload+halo %5.0f  bar1 %5.0f  tree %5.0f  bar2 %5.0f  final %5.0f  dmawait %5.0f  stagebar %5.0f  - %5.0f" % (
            tot.mean(), tot.mean(1).max(), *st.mean((0, 1)))
    if has_stamps and os.environ.get("XWAVES"):
        for wv in range(8):
            line += "\n      wave (half %d, class %d): " % (wv >> 2, wv & 3) + " ".join("%7.0f" % v for v in st[:, wv, :].mean(0))
    has6 = hasattr(L, "lic360_dc6_stamps")
    if has6:
        L.lic360_dc6_stamps.argtypes = [C.c_void_p, C.c_int]; L.lic360_dc6_stamps(None, 1)
    t4 = run(L.lic360_cconv4_dc_plane, packed4, p)
    print(line + "   | old %7.1f us" % t4)
    if has6:                                                 # -DDC6_STAMP build (tools/dc6_stamp.sh): the 4x4x1 kernel's cycles per phase and wave
        buf = (C.c_ulonglong * (256 * 12 * 10))()
        L.lic360_dc6_stamps(buf, 0)
        st = np.array(buf, dtype=np.float64).reshape(256, 12, 10) / 23.0
        m = st.mean((0, 1))
        names = ["task switch", "barrier", "half 0", "x issue", "half 1", "weights", "dsteps", "tasks", "lds write", "total"]
        print("   4x4x1 kernel, cycles per wave and launch: " + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, m)))
        print("   per double step: switch %.0f  barrier %.0f  half0 %.0f  lds write %.0f  x issue %.0f  half1 %.0f  weights %.0f  (sum %.0f); per task: switch %.0f, %.1f double steps; workgroup totals min %.0f mean %.0f max %.0f" % (
            m[0] / m[6], m[1] / m[6], m[2] / m[6], m[8] / m[6], m[3] / m[6], m[4] / m[6], m[5] / m[6], (m[:6].sum() + m[8]) / m[6], m[0] / max(m[7], 1), m[6] / max(m[7], 1),
            st[:, :, 9].mean(1).min(), st[:, :, 9].mean(), st[:, :, 9].mean(1).max()))
        if os.environ.get("XWAVES"):
            for wv in range(12): print("      wave (set %d, class %d): " % (wv >> 2, wv & 3) + " ".join("%8.0f" % v for v in st[:, wv, :].mean(0)))
    tot16 += t16; tot4 += t4
print("mean over planes: new %.1f us, old %.1f us" % (tot16 / len(planes), tot4 / len(planes)))
