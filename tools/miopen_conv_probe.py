"""what torch.nn.functional.conv2d (MIOpen) reaches on the analysis/synthesis transform's conv shapes (fp32), for sizing f1"""
import torch, time
dev = "cuda:0"
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps
for (n, cin, cout, h, w, k, s) in [(8, 192, 192, 256, 512, 3, 1), (8, 192, 192, 128, 256, 3, 1), (8, 192, 192, 260, 516, 3, 2), (8, 192, 192, 128, 256, 1, 1), (8, 96, 96, 128, 256, 3, 1), (8, 192, 768, 64, 128, 3, 1)]:
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    try:
        dt = t(lambda: torch.nn.functional.conv2d(x, wt, b, stride=s, padding=k // 2))
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        fl = 2.0 * n * cout * cin * k * k * ho * wo
        print("conv %dx%d s%d %d->%d @%dx%d x%d: %.2f ms  %.1f TFLOP/s" % (k, k, s, cin, cout, h, w, n, dt * 1e3, fl / dt / 1e12), flush=True)
    except Exception as e:
        print("conv failed", (n, cin, cout, h, w, k, s), repr(e)[:200], flush=True)
