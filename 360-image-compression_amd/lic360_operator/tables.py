"""EntropyGmmTable / EntropyBatchGmmTable / EntropyTable / EntropyGmm modules
(reference: lic360_operator/EntropyGmmTable.py:23-58, EntropyTable.py:20-29, EntropyGmm.py:22-31)."""
import torch
import lic360
from .base import BaseOpModule, contiguous


class EntropyGmmTable(BaseOpModule):
    def __init__(self, nstep, bias, num_gaussian, total_region=65536, beta=1e-6, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.EntropyGmmTableOp(nstep, bias, num_gaussian, total_region, beta, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, weight, delta, mean, ntop):
        return self._op(weight).forward(contiguous(weight), contiguous(delta), contiguous(mean), ntop)[0]


class EntropyBatchGmmTable(BaseOpModule):
    def __init__(self, nstep, bias, num_gaussian, total_region=65536, beta=1e-6, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.EntropyGmmTableOp(nstep, bias, num_gaussian, total_region, beta, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x, ntop):
        return self._op(x).forward_batch(contiguous(x), ntop)[0]


class EntropyTable(BaseOpModule):
    def __init__(self, nstep, totoal_region=65536, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.EntropyTableOp(nstep, totoal_region, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x, count):
        return self._op(x).forward(contiguous(x), count)[0]


class _EntropyGmmFn(torch.autograd.Function):
    """loss = -log(sum_i w_i (Phi_b - Phi_a) + 1e-7); analytic grads are produced by the forward kernel
    and scaled by the incoming gradient (extension/entropy_gmm_cuda.cu:36-68,94-106)."""

    @staticmethod
    def forward(ctx, weight, delta, mean, label, op):
        out = op.forward(contiguous(weight), contiguous(delta), contiguous(mean), contiguous(label))
        ctx.op = op
        return out[0]

    @staticmethod
    def backward(ctx, grad_output):
        d = ctx.op.backward(contiguous(grad_output))
        return d[0], d[1], d[2], d[3], None


class EntropyGmm(BaseOpModule):
    def __init__(self, num_gaussian=3, ignore_label=0, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.EntropyGmmOp(num_gaussian, ignore_label, gid, time_it) for gid in self.device_list}

    def forward(self, weight, delta, mean, label):
        return _EntropyGmmFn.apply(weight, delta, mean, label, self._op(weight))
