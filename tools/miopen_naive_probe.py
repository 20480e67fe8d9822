"""Which convolution of the transforms ends up on MIOpen's naive solver, and does `torch.backends.cudnn.benchmark = True` (find mode)
change the transforms' time?  Times every distinct conv shape of lic360_models' CMP_Encoder / CMP_Decoder alone (batch 8) in both modes."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "360-image-compression_amd"))
import torch, torch.nn as nn
import lic360_models as lm
dev = "cuda:0"
torch.manual_seed(0)
enc, dec = lm.CMP_Encoder(gpu_id=0).to(dev).eval(), lm.CMP_Decoder(gpu_id=0).to(dev).eval()
shapes = collections.OrderedDict()
def hook(m, inp, out):
    k = (tuple(inp[0].shape), m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding)
    shapes[k] = shapes.get(k, 0) + 1
hs = [m.register_forward_hook(hook) for net in (enc, dec) for m in net.modules() if isinstance(m, nn.Conv2d)]
with torch.no_grad():
    img = torch.rand((8, 3, 512, 1024), device=dev)
    code, mask, _ = enc(img); dec(code, mask)
for h in hs: h.remove()
def t_conv(k, reps=5):
    ishape, cin, cout, ks, st, pd = k
    conv = nn.Conv2d(cin, cout, ks, st, pd).to(dev)
    x = torch.randn(ishape, device=dev)
    with torch.no_grad():
        conv(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): y = conv(x)
        e1.record(); torch.cuda.synchronize()
    fl = 2.0 * y.numel() * cin * ks[0] * ks[1]
    return e0.elapsed_time(e1) / reps, fl
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    tot = 0.0
    print("== cudnn.benchmark =", bench)
    for k, cnt in shapes.items():
        ms, fl = t_conv(k)
        tot += ms * cnt
        print("  x%2d in %-22s %3d->%3d k%s s%s p%s : %8.3f ms  %6.1f TFLOP/s" % (cnt, k[0], k[1], k[2], k[3][0], k[4][0], k[5][0], ms, fl / ms / 1e9))
    print("  sum over both transforms (batch 8): %.1f ms = %.2f ms per image" % (tot, tot / 8))
