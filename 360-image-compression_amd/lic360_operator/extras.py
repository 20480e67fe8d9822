"""The rest of the names `lic360_operator/__init__.py:1-29` exports, so that `test/lic360_demo.py:3-4` and
`test/model_zoo.py:3` import unchanged.  They sit OUTSIDE the accelerated path (pure-torch activation /
metric / bookkeeping utilities, and two wrappers of out-of-scope native ops) and are plain PyTorch-ROCm here:

  GDN            generalized divisive normalisation, y = x / sqrt(beta + sum_j gamma_ij x_j^2) (lic360_operator/GDN.py:26-100):
                 same constructor, parameter names (`beta`, `gamma`) and reparametrisation, so checkpoints load.
  DropGrad       identity with a gradient gate (lic360_operator/DropGrad.py:4-24)
  SSIM           11x11 Gaussian-window SSIM (lic360_operator/pytorch_ssim.py:16-63)
  ModuleSaver    best / latest checkpoint writer (lic360_operator/ModuleSaver.py:4-35)
  Logger         screen + file log (lic360_operator/Logger.py:3-23)
  MultiProject   the 14 viewports of lic360.ProjectsOp (viewport metrics)
  MaskConv2      torch conv2d over a weight masked by lic360.MaskConstrainOp (the training-time form of the context conv)
"""
import math
import os

import torch
from torch import nn
import torch.nn.functional as F


class _FloorSTE(torch.autograd.Function):
    """max(x, bound) whose gradient also passes where it would move x back above the bound."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = float(bound)
        return x.clamp_min(ctx.bound)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        keep = (x >= ctx.bound) | (g < 0)
        return g * keep.to(g.dtype), None


class GDN(nn.Module):
    def __init__(self, ch, device=0, inverse=False, beta_min=1e-6, gamma_init=.1, reparam_offset=2 ** -18):
        super().__init__()
        self.inverse = bool(inverse)
        dev = torch.device("cuda:%d" % (device if isinstance(device, int) else device[0])) if torch.cuda.is_available() else torch.device("cpu")
        ped = float(reparam_offset) ** 2
        self.pedestal = ped
        self.beta_bound = math.sqrt(float(beta_min) + ped)
        self.gamma_bound = float(reparam_offset)
        self.beta = nn.Parameter(torch.sqrt(torch.ones(ch) + ped).to(dev))
        self.gamma = nn.Parameter(torch.sqrt(float(gamma_init) * torch.eye(ch) + ped).to(dev))

    def forward(self, x):
        shape = x.shape
        if x.dim() == 5:
            x = x.reshape(shape[0], shape[1], shape[2] * shape[3], shape[4])
        ch = x.shape[1]
        beta = _FloorSTE.apply(self.beta.to(x.device), self.beta_bound) ** 2 - self.pedestal
        gamma = _FloorSTE.apply(self.gamma.to(x.device), self.gamma_bound) ** 2 - self.pedestal
        if not (torch.is_grad_enabled() and (x.requires_grad or self.beta.requires_grad)) and x.is_cuda and x.dtype == torch.float32:
            import lic360
            if lic360.gdn_supported(ch):                                    # one fused pass instead of four torch kernels
                return lic360.gdn_forward(x.contiguous(), gamma.detach(), beta.detach(), self.inverse).reshape(shape)
        norm = torch.sqrt(F.conv2d(x * x, gamma.view(ch, ch, 1, 1), beta))
        y = x * norm if self.inverse else x / norm
        return y.reshape(shape)


class _GradGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep):
        ctx.keep = keep
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g * ctx.keep, None


class DropGrad(nn.Module):
    def __init__(self, drop=True):
        super().__init__()
        self.drop = 0 if drop else 1            # multiplier of the incoming gradient

    def forward(self, x):
        return _GradGate.apply(x, self.drop)


def _gauss_window(size, channel, sigma=1.5):
    k = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-k * k / (2.0 * sigma * sigma))
    g = g / g.sum()
    return (g[:, None] * g[None, :]).expand(channel, 1, size, size).contiguous()


class SSIM(nn.Module):
    def __init__(self, window_size=11, channel=1, size_average=True):
        super().__init__()
        self.window_size, self.channel, self.size_average = int(window_size), int(channel), bool(size_average)
        self.window = _gauss_window(self.window_size, self.channel)

    def forward(self, a, b):
        ch = a.shape[1]
        if ch != self.channel or self.window.device != a.device or self.window.dtype != a.dtype:
            self.window, self.channel = _gauss_window(self.window_size, ch).to(device=a.device, dtype=a.dtype), ch
        w, p = self.window, self.window_size // 2
        blur = lambda t: F.conv2d(t, w, padding=p, groups=ch)
        mu_a, mu_b = blur(a), blur(b)
        var_a, var_b, cov = blur(a * a) - mu_a * mu_a, blur(b * b) - mu_b * mu_b, blur(a * b) - mu_a * mu_b
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m = ((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (var_a + var_b + c2))
        return m.mean() if self.size_average else m.mean(dim=(1, 2, 3))


class ModuleSaver(object):
    """save(model, loss) keeps `<prex>_best_<i>.pt` per tracked loss and `<prex>_latest.pt` otherwise."""

    def __init__(self, path="./saved_models/", prex="default"):
        self.path, self.prex = path, prex
        os.makedirs(path, exist_ok=True)
        self.current_best_loss, self.init = None, False

    def init_loss(self, loss):
        self.current_best_loss = list(loss) if isinstance(loss, list) else [loss]
        self.init = True

    def save(self, model, loss):
        wrapped = isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel))
        state = (model.module if wrapped else model).state_dict()
        loss = loss if isinstance(loss, list) else [loss]
        if not self.init:
            self.init_loss([10e9] * len(loss))
        msg = ""
        for i, v in enumerate(loss):
            if v < self.current_best_loss[i]:
                self.current_best_loss[i] = v
                torch.save(state, os.path.join(self.path, "%s_best_%d.pt" % (self.prex, i)))
                msg += "save %s_best_%d.pt\t" % (self.prex, i)
        if not msg:
            torch.save(state, os.path.join(self.path, "%s_latest.pt" % self.prex))
            msg = "update %s_latest.pt" % self.prex
        return msg


class Logger(object):
    def __init__(self, fname, screen=True, file=True):
        self.screen_out, self.file = screen, file
        self.fout = open(fname, "w") if file else None

    def log(self, *args):
        if self.screen_out:
            print(*args)
        if self.fout:
            self.fout.write(" ".join(str(a) for a in args) + "\n")
            self.fout.flush()

    def close(self):
        if self.fout:
            self.fout.close()
            self.fout = None

    def __del__(self):
        self.close()


class MultiProject(nn.Module):
    """14 viewports of every ERP image of a batch -- four around the equator, four each at +-45 degrees, the two poles -- for the
    viewport metrics (reference lic360_operator/MultiProject.py:24-35; used as `MultiProject(171, 256, 0.5, False, gpu)` by
    test/lic360_demo.py:424-425).  Output [n*14, c, h, w], viewport-major."""
    THETAS = (-0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, -0.5, 0, 0.5, 1, 0, 0)
    PHIS = (0, 0, 0, 0, 0.25, 0.25, 0.25, 0.25, -0.25, -0.25, -0.25, -0.25, 0.5, -0.5)

    def __init__(self, h, w, fov=0.6, near=False, device_id=0, time_flag=False):
        super().__init__()
        import lic360
        from .autograd import UnaryFn, recording
        self._fn, self._recording = UnaryFn, recording
        self.thetas, self.phis = list(self.THETAS), list(self.PHIS)
        devs = [device_id] if isinstance(device_id, int) else list(device_id)
        self.op = {gid: lic360.ProjectsOp(int(h), int(w), self.thetas, self.phis, fov, near, gid, time_flag) for gid in devs}

    def forward(self, x):
        x = x if x.is_contiguous() else x.contiguous()
        op = self.op[x.device.index]
        if self._recording(x):
            return self._fn.apply(x, op, False)
        with torch.no_grad():
            return op.forward(x)[0]


class _MaskConstrainFn(torch.autograd.Function):
    """identity on the tensor it is given, after masking it in place; the gradient is masked the same way"""
    @staticmethod
    def forward(ctx, x, op):
        op[x.device.index].forward(x)
        ctx.op = op
        return x

    @staticmethod
    def backward(ctx, grad_output):
        ctx.op[grad_output.device.index].backward(grad_output)
        return grad_output, None


class MaskConv2(nn.Module):
    """The training-time form of the group-causal context convolution (reference lic360_operator/MaskConstrain.py:24-39): a dense
    `conv2d` whose weight is masked by lic360.MaskConstrainOp before every forward.  Same constructor, same parameter names
    (`weight` [c_out*ngroup, c_in*ngroup, k, k], `bias`), so checkpoints interchange; the inference path evaluates the same
    function with CconvEc / CconvDc."""
    def __init__(self, ngroup, c_in, c_out, kernel_size, hidden=False, device=0, time_it=False):
        super().__init__()
        import lic360
        devs = [device] if isinstance(device, int) else list(device)
        self.op = {gid: lic360.MaskConstrainOp(6 if hidden else 5, ngroup, gid, time_it) for gid in devs}
        self.weight = nn.Parameter(torch.empty((c_out * ngroup, c_in * ngroup, kernel_size, kernel_size), dtype=torch.float32))
        nn.init.kaiming_normal_(self.weight)
        self.bias = nn.Parameter(torch.zeros(c_out * ngroup, dtype=torch.float32))
        self.pad = kernel_size // 2

    def forward(self, x):
        self.weight.data = _MaskConstrainFn.apply(self.weight.data, self.op)
        return nn.functional.conv2d(x, self.weight, self.bias, padding=self.pad)
