"""lic360_cconv4_dc_plane alone: one decode-order layer (default: a hidden layer), N samples (default 144 = 48 images x 3 nets), chosen
planes; per-launch time from HIP events and -- in a diagnostic build with -DDC6_STAMP (tools/dc6_stamp.sh) -- the cycles each wave spent
per phase.  XP=planes, XN=samples, XCOUT=3 (last layer), XCIN=1 (first layer), XCOLD=k (rotate over k input buffers), XWAVES=1 (per wave)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("360-image-compression_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import lic360 as lic
from util import conv_params
G, H, W, N = 48, 64, 128, int(os.environ.get("XN", 144))
CIN, COUT = int(os.environ.get("XCIN", 4)), int(os.environ.get("XCOUT", 4))
planes = [int(v) for v in os.environ.get("XP", "30,70,110,150,190,220").split(",")]
L = lic._lib
plan = C.c_void_p(0)
assert L.lic360_conv_plan_create(G * CIN, G, G * COUT, 5, 6 if CIN == 4 else 5, C.byref(plan)) == 0
rng = np.random.default_rng(0)
w, b, a = conv_params(rng, 3, G * COUT, G * CIN, act=True)
wd, bd, ad = (torch.from_numpy(t).cuda() for t in (w, b, a))
s = lic._stream(0); P = lic._p
nfl_in, nfl_out = L.lic360_conv4_buffer_floats(0, N * G * CIN, H, W), L.lic360_conv4_buffer_floats(0, N * G * COUT, H, W)
NX = int(os.environ.get("XCOLD", 1))                    # > 1: rotate over this many input buffers, so that every launch reads cold activations
xs = [torch.randn(nfl_in, dtype=torch.float32, device="cuda:0") for _ in range(NX)]
out = torch.zeros(nfl_out, dtype=torch.float32, device="cuda:0")
packed4 = torch.empty(3 * L.lic360_conv4_packed_floats(plan), dtype=torch.float32, device="cuda:0")
assert L.lic360_conv4_pack(s, plan, P(wd), 3, P(packed4)) == 0
def run(fn, pk, p, reps=20):
    for i in range(3): assert fn(s, plan, P(xs[i % NX]), P(pk), P(bd), P(ad), None, P(out), N, H, W, 3, p, N) == 0, L.lic360_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(s, plan, P(xs[i % NX]), P(pk), P(bd), P(ad), None, P(out), N, H, W, 3, p, N)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
tot4 = 0.0
has6 = hasattr(L, "lic360_dc6_stamps")
for p in planes:
    if has6:
        L.lic360_dc6_stamps.argtypes = [C.c_void_p, C.c_int]; L.lic360_dc6_stamps(None, 1)
    t4 = run(L.lic360_cconv4_dc_plane, packed4, p)
    print("plane %3d: %7.1f us" % (p, t4))
    if has6:                                                 # -DDC6_STAMP build (tools/dc6_stamp.sh): the 4x4x1 kernel's cycles per phase and wave
        buf = (C.c_ulonglong * (256 * 12 * 10))()
        L.lic360_dc6_stamps(buf, 0)
        st = np.array(buf, dtype=np.float64).reshape(256, 12, 10) / 23.0
        m = st.mean((0, 1))
        names = ["task switch", "barrier", "half 0", "x issue", "half 1", "weights", "dsteps", "tasks", "lds write", "total"]
        print("   cycles per wave and launch: " + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, m)))
        print("   per double step: switch %.0f  barrier %.0f  half0 %.0f  lds write %.0f  x issue %.0f  half1 %.0f  weights %.0f  (sum %.0f); per task: switch %.0f, %.1f double steps; workgroup totals min %.0f mean %.0f max %.0f" % (
            m[0] / m[6], m[1] / m[6], m[2] / m[6], m[8] / m[6], m[3] / m[6], m[4] / m[6], m[5] / m[6], (m[:6].sum() + m[8]) / m[6], m[0] / max(m[7], 1), m[6] / max(m[7], 1),
            st[:, :, 9].mean(1).min(), st[:, :, 9].mean(), st[:, :, 9].mean(1).max()))
        if os.environ.get("XWAVES"):
            for wv in range(12): print("      wave (set %d, class %d): " % (wv >> 2, wv & 3) + " ".join("%8.0f" % v for v in st[:, wv, :].mean(0)))
    tot4 += t4
print("mean over planes: %.1f us" % (tot4 / len(planes)))
