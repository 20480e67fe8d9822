"""lic360_sconv3x3 (csrc/conv3x3_kernels.hip) next to torch.nn.functional.conv2d (MIOpen) on the transforms' 3x3 stride-1 shapes, batch 8:
correctness (max |diff| against conv2d over the sphere-padded map, fp32) and time per launch -- the library form timed WITH the launches the
fused kernel absorbs (in-place apron refresh, PReLU, trim) and without.  VERDICT r4 next #2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "360-image-compression_amd"))
import torch
import torch.nn.functional as F
import lic360

dev = "cuda:0"


def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
# ring 1 = the first convolution of ResidualBlockV2: rows of the 1-ring window, columns of the interior (its 1-ring columns repeat interior columns: ring_w = 2)
shapes = [(8, 192, 192, 260, 516, 1, 0), (8, 192, 192, 260, 516, 2, 0), (8, 192, 192, 132, 260, 1, 0), (8, 192, 192, 132, 260, 2, 0),
          (8, 96, 96, 132, 260, 2, 0), (8, 192, 768, 68, 132, 2, 1), (8, 192, 768, 132, 260, 2, 1), (8, 192, 768, 36, 68, 2, 1), (8, 192, 192, 36, 68, 2, 0)]
if os.environ.get("S3_QUICK"): shapes = shapes[2:3] + shapes[6:7]
if os.environ.get("S3_N"): shapes = [(int(os.environ["S3_N"]),) + sh[1:] for sh in shapes]      # another batch size (default 8)
for (n, cin, cout, hp, wp, ring, crop) in shapes:
    x = torch.randn(n, cin, hp, wp, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b, sl = torch.randn(cout, device=dev), torch.rand(cout, device=dev) * 0.5
    res = torch.randn(n, cout, hp, wp, device=dev) if crop == 0 else None
    pk = lic360.sconv3x3_pack(w)
    out = torch.zeros(n, cout, hp - 2 * crop, wp - 2 * crop, device=dev)
    pad_op, trim_op = lic360.SpherePadOp(2, True, 0, False), lic360.SphereTrimOp(ring, 0, False)

    def lib(full):
        xp = pad_op.forward(x)[0] if full else x                       # in-place apron refresh
        y = F.conv2d(xp, w, b, padding=1 - crop)
        if full:
            y = F.prelu(y, sl)
            if crop == 0:
                y = trim_op.forward(y)[0]
                y = y + res
        return y

    def ours():
        return lic360.sconv3x3(x, pk, b, sl, res, out, pad=2, sphere=True, ring=ring, crop=crop, ring_w=max(ring, 2 - 2 * crop))

    want = lib(True)                                                    # (refreshes x's apron in place: the fused kernel ignores it anyway)
    got = ours()
    if crop == 0:
        win = (slice(None), slice(None), slice(ring, hp - ring), slice(2, wp - 2))
        err = (got[win] - want[win]).abs().max().item()
    else:                                                               # unpadded conv: out cell (i, j) = input cell (i + 1, j + 1); window in out coords
        win = (slice(None), slice(None), slice(ring - crop, hp - crop - ring), slice(ring - crop, wp - crop - ring))
        err = (got[win] - want[win]).abs().max().item()
    t_conv, t_full, t_ours = timed(lambda: lib(False)), timed(lambda: lib(True)), timed(ours)
    fl_lib = 2.0 * n * cout * cin * 9 * (hp - 2 * crop) * (wp - 2 * crop)
    rw_ = max(ring, 2 - 2 * crop)
    fl_win = 2.0 * n * cout * cin * 9 * (hp - 2 * ring) * (wp - 2 * rw_)
    print("3x3 %d->%d @%dx%d x%d ring %d crop %d: max|diff| %.2e | MIOpen conv alone %.3f ms (%.1f TF), with pad+prelu+trim+add %.3f ms | sconv3x3 %.3f ms "
          "(%.1f TF on its %dx%d window, %.1f TF nominal) | ours / library-with-epilogue = %.2f" % (
              cin, cout, hp, wp, n, ring, crop, err, t_conv, fl_lib / t_conv / 1e9, t_full, t_ours, fl_win / t_ours / 1e9, hp - 2 * ring, wp - 2 * rw_,
              fl_lib / t_ours / 1e9, t_ours / t_full), flush=True)

# ---- the 1x1 layers of ResidualBlock on the same kernel body, against the library GEMM (+ the elementwise launch each one absorbs)
for (n, cin, cout, hp, wp, what) in [(8, 192, 96, 132, 260, "prelu"), (8, 96, 192, 132, 260, "add"), (8, 192, 192, 132, 260, "prelu")]:
    if os.environ.get("S3_N"): n = int(os.environ["S3_N"])
    x = torch.randn(n, cin, hp, wp, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    b, sl = torch.randn(cout, device=dev), torch.rand(cout, device=dev) * 0.5
    res = torch.randn(n, cout, hp, wp, device=dev)
    pk = lic360.sconv1x1_pack(w)
    out = torch.zeros(n, cout, hp, wp, device=dev)
    lib = (lambda: F.prelu(F.conv2d(x, w, b), sl)) if what == "prelu" else (lambda: F.conv2d(x, w, b) + res)
    ours = (lambda: lic360.sconv1x1(x, pk, b, sl, None, out, ring=2)) if what == "prelu" else (lambda: lic360.sconv1x1(x, pk, b, None, res, out, ring=2))
    win = (slice(None), slice(None), slice(2, hp - 2), slice(2, wp - 2))
    err = (ours()[win] - lib()[win]).abs().max().item()
    t_conv, t_lib, t_ours = timed(lambda: F.conv2d(x, w, b)), timed(lib), timed(ours)
    fl = 2.0 * n * cout * cin * (hp - 4) * (wp - 4)
    print("1x1 %d->%d @%dx%d x%d + %s: max|diff| %.2e | library conv alone %.3f ms, with the %s %.3f ms | sconv1x1 %.3f ms (%.1f TF on its window) | ours / library = %.2f" % (
        cin, cout, hp, wp, n, what, err, t_conv, what, t_lib, t_ours, fl / t_ours / 1e9, t_ours / t_lib), flush=True)
