"""CodeContex, TileExtract(Batch), TileInput, TileAdd -- scan order and plane gather/scatter modules
(reference: lic360_operator/CodeContex.py:21-30, TileExtract.py:19-69, TileInput.py:21-37, TileAdd.py:19-36)."""
import torch
import lic360
from .base import BaseOpModule, contiguous


class CodeContex(BaseOpModule):
    def __init__(self, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.CodeContexOp(gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x):
        out = self._op(x).forward(x)
        return out[0], out[1]


class TileExtract(BaseOpModule):
    def __init__(self, ngroup, label, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.TileExtractOp(ngroup, label, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x):
        out = self._op(x).forward(contiguous(x))
        return out[0], out[1]


class TileExtractBatch(BaseOpModule):
    def __init__(self, ngroup, label, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.TileExtractOp(ngroup, label, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x):
        out = self._op(x).forward_batch(contiguous(x))
        return out[0], out[1]


class TileInput(BaseOpModule):
    def __init__(self, ngroup, bias=0., scale=1., replicate=1, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.TileInputOp(ngroup, bias, scale, replicate, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x):
        return self._op(x).forward(contiguous(x))[0]


class TileAdd(BaseOpModule):
    def __init__(self, ngroup, device=0, time_it=False):
        super().__init__(device)
        self.op = {gid: lic360.TileAddOp(ngroup, gid, time_it) for gid in self.device_list}

    @torch.no_grad()
    def forward(self, x, y):
        return self._op(x).forward(x, y)[0]
