"""CconvEc / CconvEcBatch / CconvDc / CconvDcBatch -- group-causal masked 5x5 convolution layers of
the entropy nets (reference: lic360_operator/CconvEc.py:62-97, lic360_operator/CconvDc.py:58-112).
Parameter names (weight / bias / relu) and shapes match the reference so that
cast_entropy_parameter (test/lic360_demo.py:296-322) can fill them."""
import torch
import torch.nn as nn
import lic360
from .base import BaseOpModule, contiguous


class _Cconv(BaseOpModule):
    OP = None

    def __init__(self, ngroup, c_in, c_out, kernel_size, batch, hidden, act, device, time_it):
        super().__init__(device)
        constrain = 6 if hidden else 5              # CconvEc.py:64
        channel, nout = ngroup * c_in, ngroup * c_out
        self.op = {gid: self.OP(channel, ngroup, nout, kernel_size, constrain, gid, time_it) for gid in self.device_list}
        lead = () if batch is None else (batch,)
        if batch is None:
            self.weight = nn.Parameter(torch.empty(lead + (nout, channel, kernel_size, kernel_size), dtype=torch.float32))
            nn.init.kaiming_normal_(self.weight)
            self.bias = nn.Parameter(torch.zeros(lead + (nout,), dtype=torch.float32))
            self.relu = nn.Parameter(torch.zeros(lead + (nout,), dtype=torch.float32)) if act else None
        else:
            self.weight = nn.Parameter(torch.rand(lead + (nout, channel, kernel_size, kernel_size), dtype=torch.float32))
            self.bias = nn.Parameter(torch.rand(lead + (nout,), dtype=torch.float32))
            self.relu = nn.Parameter(torch.rand(lead + (nout,), dtype=torch.float32)) if act else None
        self.act = act
        self.batch = batch

    @torch.no_grad()
    def forward(self, x):
        op = self._op(x)
        x = contiguous(x)
        if self.batch is None:
            out = op.forward_act(x, self.weight, self.bias, self.relu) if self.act else op.forward(x, self.weight, self.bias)
        else:
            out = op.forward_act_batch(x, self.weight, self.bias, self.relu) if self.act else op.forward_batch(x, self.weight, self.bias)
        return out[0]


class CconvEc(_Cconv):
    OP = lic360.CconvEcOp

    def __init__(self, ngroup, c_in, c_out, kernel_size, hidden=False, act=True, device=0, time_it=False):
        super().__init__(ngroup, c_in, c_out, kernel_size, None, hidden, act, device, time_it)


class CconvEcBatch(_Cconv):
    OP = lic360.CconvEcOp

    def __init__(self, ngroup, c_in, c_out, kernel_size, batch=3, hidden=False, act=True, device=0, time_it=False):
        super().__init__(ngroup, c_in, c_out, kernel_size, batch, hidden, act, device, time_it)


class CconvDc(_Cconv):
    OP = lic360.CconvDcOp

    def __init__(self, ngroup, c_in, c_out, kernel_size, hidden=False, act=True, device=0, time_it=False):
        super().__init__(ngroup, c_in, c_out, kernel_size, None, hidden, act, device, time_it)


class CconvDcBatch(_Cconv):
    OP = lic360.CconvDcOp

    def __init__(self, ngroup, c_in, c_out, kernel_size, batch=3, hidden=False, act=True, device=0, time_it=False):
        super().__init__(ngroup, c_in, c_out, kernel_size, batch, hidden, act, device, time_it)
