"""BaseOpModule -- nn.Module shell that owns one native op object per device
(counterpart of lic360_operator/BaseOpModule.py:5-54 in the reference)."""
import torch
from torch import nn


class BaseOpModule(nn.Module):
    def __init__(self, devices=0):
        super().__init__()
        self.device_list = [devices] if isinstance(devices, int) else list(devices)
        self.op = {}

    def _rekey(self, device):
        """Move the single native op to `device` (reference: custom_op_to, BaseOpModule.py:33-40)."""
        if device is None or device.type != "cuda" or len(self.op) != 1:
            return
        new_id = device.index if device.index is not None else torch.cuda.current_device()
        old_id = next(iter(self.op))
        if new_id != old_id:
            self.op[new_id] = self.op.pop(old_id)
            self.op[new_id].to(new_id)
            self.device_list = [new_id]

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = torch.device(a)
            elif isinstance(a, int):
                device = torch.device("cuda", a)
        if isinstance(device, (str, int)):
            device = torch.device(device) if isinstance(device, str) else torch.device("cuda", device)
        out = super().to(*args, **kwargs)
        for m in self.modules():
            if isinstance(m, BaseOpModule):
                m._rekey(device)
        return out

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def _op(self, x):
        gid = x.device.index
        if gid not in self.op:
            self._rekey(x.device)
        return self.op[gid]

    def restart(self):
        for op in self.op.values():
            if hasattr(op, "restart"):
                op.restart()

    def set_param(self, p1, p2):
        for gid, op in self.op.items():
            op.set_param(p1.to("cuda:{}".format(gid)), p2)


def contiguous(x):
    return x if x.is_contiguous() else x.contiguous()
