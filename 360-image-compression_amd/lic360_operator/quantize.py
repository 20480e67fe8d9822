"""ImpMap / QUANT / Dquant / Dtow / Imp2mask / Scale / ContextReshape / ContextShift modules
(reference: lic360_operator/ImpMap.py:59-75, QUANT.py:31-45, Dquant.py:21-31, Dtow.py:22-32,
Imp2mask.py:19-28, Scale.py:22-31, ContextReshape.py:22-29, ContextShift.py:22-30)."""
import math
import torch
import torch.nn as nn
import lic360
from .base import BaseOpModule, contiguous
from .autograd import UnaryFn, ImpMapFn, QuantFn, recording


class ImpMap(BaseOpModule):
    def __init__(self, rt, alpha, gamma, levels, scale_constrain=1., scale_weight=1., imp_kernel=0, device=0, ntop=1, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.ImpMapOp(levels, alpha, gamma, rt, scale_constrain, scale_weight, imp_kernel, ntop, gid, time_it)
                   for gid in self.device_list}
        self.level = levels
        self.ntop = ntop

    def forward(self, x, imp):
        if recording(x, imp):
            return ImpMapFn.apply(contiguous(x), imp, self.level, self._op(x), self.ntop > 1)
        with torch.no_grad():
            imp = (torch.floor(imp * self.level) / self.level).contiguous()  # the map takes `levels` + 1 values (ImpMap.py:39)
            out = self._op(x).forward(contiguous(x), imp)
            rt = torch.mean(imp)
            if self.ntop > 1:
                return out[0], out[2], rt
            return out[0], rt


class QUANT(BaseOpModule):
    def __init__(self, channel, bin_num, check_iters=100, weight_decay=0.9, ntop=1, top_alpha=0.1, device_id=0, time_flag=False):
        super().__init__(device_id)
        ta = 1. / (bin_num + 1)
        w = torch.full((channel, bin_num), math.log(ta), dtype=torch.float32)
        w[:, 0] = ta
        self.weight = nn.Parameter(w)
        self.count = nn.Parameter(torch.zeros((channel, bin_num), dtype=torch.float32))
        self.op = {gid: lic360.QuantOp(channel, bin_num, weight_decay, check_iters, ntop, top_alpha, gid, time_flag) for gid in self.device_list}

    def forward(self, x):
        if recording(x, self.weight):
            return QuantFn.apply(contiguous(x), self.weight, self.count, self._op(x), self.training)
        with torch.no_grad():
            out = self._op(x).forward(contiguous(x), self.weight, self.count, self.training)
            return out[0] if len(out) == 1 else (out[0], out[1])


class Dquant(BaseOpModule):
    def __init__(self, channel, bin_num, device=0, time_it=False):
        super().__init__(device)
        self.weight = nn.Parameter(torch.zeros((channel, bin_num), dtype=torch.float32))
        self.op = {gid: lic360.DquantOp(channel, bin_num, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x, mask):
        return self._op(x).forward(contiguous(x), contiguous(mask), self.weight)[0]


class Dtow(BaseOpModule):
    def __init__(self, stride=2, d2w=False, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.DtowOp(stride, d2w, gid, time_it) for gid in self.device_list}

    def forward(self, x):
        if recording(x):
            return UnaryFn.apply(contiguous(x), self._op(x), False)
        with torch.no_grad():
            return self._op(x).forward(contiguous(x))[0]


class Imp2mask(BaseOpModule):
    def __init__(self, levels, channels, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.Imp2maskOp(levels, channels, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x):
        return self._op(x).forward(contiguous(x))[0]


class Scale(BaseOpModule):
    def __init__(self, bias, scale, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.ScaleOp(bias, scale, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x):
        return self._op(x).forward(contiguous(x))[0]


class ContextReshape(BaseOpModule):
    def __init__(self, ngroup, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.ContextReshapeOp(ngroup, gid, time_it) for gid in self.device_list}

    def forward(self, x):
        if recording(x):
            return UnaryFn.apply(contiguous(x), self._op(x), False)
        with torch.no_grad():
            return self._op(x).forward(contiguous(x))[0]


class ContextShift(BaseOpModule):
    def __init__(self, inv, cpn=1, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.ContexShiftOp(inv, cpn, gid, time_it) for gid in self.device_list}

    def forward(self, x):
        if recording(x):
            return UnaryFn.apply(contiguous(x), self._op(x), False)
        with torch.no_grad():
            return self._op(x).forward(contiguous(x))[0]
