"""SpherePad / SphereTrim / SphereCutEdge / SphereLatScaleNet -- ERP border modules
(reference: lic360_operator/SpherePad.py:24-34, SphereTrim.py:24-31, SphereCutEdge.py:24-33,
SphereLatScaleNet.py:40-62)."""
import numpy as np
import torch
import torch.nn as nn
import lic360
from .base import BaseOpModule, contiguous
from .autograd import UnaryFn, LatScaleFn, recording


class SpherePad(BaseOpModule):
    def __init__(self, pad, device=0, inplace=False, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.SpherePadOp(pad, inplace, gid, time_it) for gid in self.device_list}
        self.inplace = bool(inplace)

    def forward(self, x):
        if recording(x):
            return UnaryFn.apply(contiguous(x), self._op(x), self.inplace)
        with torch.no_grad():
            return self._op(x).forward(contiguous(x))[0]


class SphereTrim(BaseOpModule):
    def __init__(self, pad, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.SphereTrimOp(pad, gid, time_it) for gid in self.device_list}

    def forward(self, x):
        if recording(x):
            return UnaryFn.apply(x, self._op(x), True)
        with torch.no_grad():
            return self._op(x).forward(x)[0]


class SphereCutEdge(BaseOpModule):
    def __init__(self, pad, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.SphereCutEdgeOp(pad, gid, time_it) for gid in self.device_list}

    def forward(self, x):
        if recording(x):
            return UnaryFn.apply(contiguous(x), self._op(x), False)
        with torch.no_grad():
            return self._op(x).forward(contiguous(x))[0]


class _ScaleResidualBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.net = nn.Sequential(nn.Conv1d(channels, channels, 3, 1, 1), nn.PReLU(channels),
                                 nn.Conv1d(channels, channels, 3, 1, 1), nn.PReLU(channels))

    def forward(self, x):
        return self.net(x) + x


ScaleResidualBlock = _ScaleResidualBlock


class SphereLatScaleNet(BaseOpModule):
    """Per-latitude-band scale predicted by a tiny 1-D CNN from |cos(lat)|; keeps the reference's
    state_dict keys `net.*` and `data` (SphereLatScaleNet.py:45-57)."""

    def __init__(self, npart, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.SphereLatScaleOp(npart, gid, time_it) for gid in self.device_list}
        self.net = nn.Sequential(nn.Conv1d(1, 16, 3, 1, 1), nn.PReLU(16), _ScaleResidualBlock(16), _ScaleResidualBlock(16),
                                 nn.Conv1d(16, 1, 1, 1), nn.Sigmoid())
        self.net[4].bias.data.fill_(3)
        ct = np.fabs(np.cos((0.5 - (np.arange(npart) + 0.5) / npart) * np.pi))
        ct = ct / np.max(ct)
        self.data = nn.Parameter(torch.from_numpy(ct).type(torch.float32).view(1, 1, npart), requires_grad=False)

    def forward(self, x):
        if recording(x, *self.net.parameters()):
            weight = self.net(self.data.data).contiguous()
            return LatScaleFn.apply(contiguous(x), weight, self._op(x))
        with torch.no_grad():
            weight = self.net(self.data.data).contiguous()
            return self._op(x).forward(contiguous(x), weight)[0]
