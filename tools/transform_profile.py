"""analysis + synthesis transforms alone (batch 8, seeded weights), a few passes: run under `rocprofv3 --kernel-trace --stats` to see which
kernels the transforms' time goes to (lic360.sconv3x3 vs the MIOpen convolutions that remain, GDN, the sphere / shuffle passes).  The first
pass runs MIOpen's find-mode search (seconds of `naive_conv_*` kernels on a fresh box): tools/trace_after_marker.py sums the per-dispatch trace
BEHIND the marker kernel (`spin_kernel`, launched after the warm-up pass) instead of reading the whole-run statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "360-image-compression_amd"))
import torch
import lic360_models as lm
torch.manual_seed(0)
dev = "cuda:0"
enc, dec = lm.CMP_Encoder(gpu_id=0).to(dev).eval(), lm.CMP_Decoder(gpu_id=0).to(dev).eval()
reps = int(os.environ.get("REPS", 3))
with torch.no_grad():
    img = torch.rand((8, 3, 512, 1024), device=dev)
    code, mask, _ = enc(img)
    dec(code, mask)
    torch.cuda.synchronize()
    torch.cuda._sleep(100000)                                # marker kernel in the trace: everything before it is warm-up (MIOpen's find-mode search)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(reps): enc(img)
    e1.record()
    for _ in range(reps): dec(code, mask)
    e2.record(); torch.cuda.synchronize()
print("analysis %.2f ms per image, synthesis %.2f ms per image" % (e0.elapsed_time(e1) / reps / 8, e1.elapsed_time(e2) / reps / 8))
