"""Codec drivers of the hot path on top of the drop-in operator surface -- counterparts of the
reference's EntEncoderFast / ImpEntEncoderFast / EntDecoder / ImpEntDecoder and the checkpoint key
remaps (test/lic360_demo.py:21-322).  Same class names, constructor arguments, sub-module names
(`net.N.weight/bias/relu`, `net.N.conv1.*`) and call protocol (`start(code_name)` then `forward`).

These drivers keep the reference's per-plane structure (one table build + one host coder call per
anti-diagonal plane).  The throughput path is lic360_fused.FusedCodec, which produces identical
bitstreams with the whole loop resident on the GPU.
"""
import torch
import lic360
from lic360_operator import (TileAdd, TileExtract, TileExtractBatch, TileInput, CodeContex, CconvDcBatch,
                             EntropyBatchGmmTable, CconvEcBatch, CconvDc, CconvEc, EntropyTable, Scale, Imp2mask, Dtow)

_PLANE_OPS = (CconvDcBatch, CconvDc, TileAdd, TileInput, TileExtract, TileExtractBatch)


class EntropyResidualBlockDBT(torch.nn.Module):          # lic360_demo.py:21-31
    def __init__(self, batch, ngroups, cpn, device_id=0):
        super().__init__()
        self.conv1 = CconvDcBatch(ngroups, cpn, cpn, 5, batch, True, True, device=device_id)
        self.conv2 = CconvDcBatch(ngroups, cpn, cpn, 5, batch, True, True, device=device_id)
        self.add = TileAdd(ngroups, device=device_id)

    def forward(self, x):
        return self.add(self.conv2(self.conv1(x)), x)


class EntropyResidualBlockDBTFast(torch.nn.Module):      # lic360_demo.py:33-41
    def __init__(self, batch, ngroups, cpn, device_id=0):
        super().__init__()
        self.conv1 = CconvEcBatch(ngroups, cpn, cpn, 5, batch, True, True, device=device_id)
        self.conv2 = CconvEcBatch(ngroups, cpn, cpn, 5, batch, True, True, device=device_id)

    def forward(self, x):
        return self.conv2(self.conv1(x)) + x


class EntropyResidualBlockD(torch.nn.Module):            # lic360_demo.py:43-53
    def __init__(self, ngroups, cpn, device_id=0):
        super().__init__()
        self.conv1 = CconvDc(ngroups, cpn, cpn, 5, True, True, device=device_id)
        self.conv2 = CconvDc(ngroups, cpn, cpn, 5, True, True, device=device_id)
        self.add = TileAdd(ngroups, device=device_id)

    def forward(self, x):
        return self.add(self.conv2(self.conv1(x)), x)


class EntropyResidualBlockDFast(torch.nn.Module):        # lic360_demo.py:55-63
    def __init__(self, ngroups, cpn, device_id=0):
        super().__init__()
        self.conv1 = CconvEc(ngroups, cpn, cpn, 5, True, True, device=device_id)
        self.conv2 = CconvEc(ngroups, cpn, cpn, 5, True, True, device=device_id)

    def forward(self, x):
        return self.conv2(self.conv1(x)) + x


def restart_entropy_network(m):
    if isinstance(m, _PLANE_OPS):
        m.restart()


def init_entropy_network(m, p1, p2):
    if isinstance(m, _PLANE_OPS):
        m.set_param(p1, p2)


class _Driver(torch.nn.Module):
    def _begin(self, ref_tensor):
        self.p1, self.p2 = self.ctx(ref_tensor)
        self.apply(lambda m: init_entropy_network(m, self.p1, self.p2))


class EntEncoderFast(_Driver):                            # lic360_demo.py:95-141
    def __init__(self, ngroup, bin_num=8, gid=0):
        super().__init__()
        self.ngroup = ngroup
        self.cuda_name = "cuda:{}".format(gid)
        self.ctx = CodeContex(device=gid)
        self.mcoder = lic360.Coder("tmp", 3.5)
        self.bias = (bin_num - 1) / 2.
        self.bin_num = bin_num
        self.net = torch.nn.Sequential(
            CconvEcBatch(ngroup, 1, 4, 5, 3, False, True, device=gid),
            *[EntropyResidualBlockDBTFast(3, ngroup, 4, gid) for _ in range(5)],
            CconvEcBatch(ngroup, 4, 3, 5, 3, True, False, device=gid))
        self.ext = TileExtractBatch(ngroup, True, device=gid)
        self.ext_label = TileExtract(ngroup, True, device=gid)
        self.ext_mask = TileExtract(ngroup, True, device=gid)
        self.gmm = EntropyBatchGmmTable(bin_num, self.bias, 3, 65536, device=gid)
        self.net = self.net.to(self.cuda_name)

    def start(self, code_name="./tmp/data"):
        self.apply(restart_entropy_network)
        self.mcoder.reset_fname(code_name)
        self.mcoder.start_encoder()

    @torch.no_grad()
    def forward(self, data, mask):
        self._begin(data)
        h, w = data.shape[2:]
        tdata = ((data - self.bias) * mask).contiguous()
        y = self.net(torch.cat([tdata, tdata, tdata], dim=0).contiguous())
        for _ in range(h + w + self.ngroup - 2):
            z, le = self.ext(y)
            vec = self.gmm(z, le)
            ln = int(le[0].item())
            label, _ = self.ext_label(data)
            tm, _ = self.ext_mask(mask)
            pred, tlabel, tm = vec.type(torch.int32).to("cpu"), label.type(torch.int32).to("cpu"), tm.type(torch.float32).to("cpu").contiguous()
            self.mcoder.encodes_mask(pred.view(-1, self.bin_num + 1), self.bin_num, tlabel.view(-1), tm.view(-1), ln)
        self.mcoder.end_encoder()


class ImpEntEncoderFast(_Driver):                         # lic360_demo.py:143-189
    def __init__(self, bin_num=48, gid=0):
        super().__init__()
        self.ngroup = 1
        cpg = bin_num * 3
        self.nsym = bin_num + 1
        self.cuda_name = "cuda:{}".format(gid)
        self.ctx = CodeContex(device=gid)
        self.mcoder = lic360.Coder("tmp", 3.5)
        self.net = torch.nn.Sequential(
            CconvEc(1, 1, cpg, 5, False, True, device=gid),
            *[EntropyResidualBlockDFast(1, cpg, gid) for _ in range(5)],
            CconvEc(1, cpg, bin_num + 1, 5, True, False, device=gid))
        self.ext = TileExtract(1, True, device=gid)
        self.ext_label = TileExtract(1, True, device=gid)
        self.table = EntropyTable(bin_num + 1, 65536, device=gid)
        self.scale = Scale(-1, float(2. / (bin_num - 1.)), device=gid)
        self.net = self.net.to(self.cuda_name)

    def start(self, code_name="./tmp/data"):
        self.apply(restart_entropy_network)
        self.mcoder.reset_fname(code_name)
        self.mcoder.start_encoder()

    @torch.no_grad()
    def forward(self, data):
        data = data.contiguous()
        self._begin(data)
        h, w = data.shape[2:]
        y = self.net(self.scale(data))
        for _ in range(h + w + self.ngroup - 2):
            z, le = self.ext(y)
            vec = self.table(z, le)
            ln = int(le[0].item())
            label, _ = self.ext_label(data)
            pred, tlabel = vec.view(-1, self.nsym + 1).type(torch.int32).to("cpu"), label.view(-1).type(torch.int32).to("cpu")
            self.mcoder.encodes(pred, self.nsym, tlabel, ln)
        self.mcoder.end_encoder()


class EntDecoder(_Driver):                                # lic360_demo.py:191-238
    def __init__(self, ngroup, bin_num=8, gid=0):
        super().__init__()
        self.cuda_name = "cuda:{}".format(gid)
        self.ctx = CodeContex(device=gid)
        self.ipt = TileInput(ngroup, -3.5, 1, 3, device=gid)
        self.ngroup = ngroup
        self.bin_num = bin_num
        self.mcoder = lic360.Coder("tmp", 3.5)
        self.bias = (bin_num - 1) / 2.
        self.net = torch.nn.Sequential(
            CconvDcBatch(ngroup, 1, 4, 5, 3, False, True, device=gid),
            *[EntropyResidualBlockDBT(3, ngroup, 4, gid) for _ in range(5)],
            CconvDcBatch(ngroup, 4, 3, 5, 3, True, False, device=gid))
        self.ext = TileExtractBatch(ngroup, True, device=gid)
        self.ext_mask = TileExtract(ngroup, True, device=gid)
        self.gmm = EntropyBatchGmmTable(bin_num, self.bias, 3, 65536, device=gid)
        self.net = self.net.to(self.cuda_name)

    def start(self, code_name="./tmp/data"):
        self.apply(restart_entropy_network)
        self.mcoder.reset_fname(code_name)
        self.mcoder.start_decoder()

    @torch.no_grad()
    def forward(self, mask):
        h, w = mask.shape[2:]
        pout = torch.zeros((1, 1, h, w), dtype=torch.float32, device=self.cuda_name)
        self._begin(pout)
        for _ in range(h + w + self.ngroup - 2):
            y = self.net(self.ipt(pout))
            z, le = self.ext(y)
            vec = self.gmm(z, le)
            ln = int(le[0].item())
            mt, _ = self.ext_mask(mask)
            pred, mt = vec.type(torch.int32).to("cpu").view(-1, self.bin_num + 1), mt.to("cpu").contiguous().view(-1)
            pout = self.mcoder.decodes_mask(pred, self.bin_num, mt, ln).to(self.cuda_name).view(1, 1, h, w).contiguous()
        b = self.ipt(pout)
        return (b[0:1] + self.bias * mask).contiguous()


class ImpEntDecoder(_Driver):                             # lic360_demo.py:241-290
    def __init__(self, bin_num=48, gid=0):
        super().__init__()
        self.ngroup = 1
        cpg = bin_num * 3
        self.nsym = bin_num + 1
        self.scale = float(2. / (bin_num - 1))
        self.cuda_name = "cuda:{}".format(gid)
        self.ctx = CodeContex(device=gid)
        self.ipt = TileInput(1, -1, self.scale, device=gid)
        self.mcoder = lic360.Coder("tmp", 3.5)
        self.i2m = Imp2mask(bin_num, bin_num * 4, gid)
        self.d2w = Dtow(2, True, gid)
        self.net = torch.nn.Sequential(
            CconvDc(1, 1, cpg, 5, False, True, device=gid),
            *[EntropyResidualBlockD(1, cpg, gid) for _ in range(5)],
            CconvDc(1, cpg, bin_num + 1, 5, True, False, device=gid))
        self.ext = TileExtract(1, True, device=gid)
        self.table = EntropyTable(bin_num + 1, 65536, device=gid)
        self.net = self.net.to(self.cuda_name)

    def start(self, code_name="./tmp/data"):
        self.apply(restart_entropy_network)
        self.mcoder.reset_fname(code_name)
        self.mcoder.start_decoder()

    @torch.no_grad()
    def forward(self, h=32, w=64):
        pout = torch.zeros((1, 1, h, w), dtype=torch.float32, device=self.cuda_name)
        self._begin(pout)
        for _ in range(h + w + self.ngroup - 2):
            y = self.net(self.ipt(pout))
            z, le = self.ext(y)
            vec = self.table(z, le)
            ln = int(le[0].item())
            pred = vec.type(torch.int32).to("cpu").view(-1, self.nsym + 1)
            pout = self.mcoder.decodes(pred, self.nsym, ln).to(self.cuda_name).view(1, 1, h, w).contiguous()
        b = self.ipt(pout)
        code = ((b + 1) / self.scale).contiguous()
        tcode = torch.floor(code + 1e-5).type(torch.float32).contiguous()
        self.last_levels = tcode
        return self.d2w(self.i2m(tcode))


def _key_map(prefix):
    """entropy-net checkpoint keys -> driver keys (lic360_demo.py:296-303)."""
    m = {"net.0.weight": "{}.0.weight", "net.0.bias": "{}.0.bias", "net.0.relu": "{}.1.weight",
         "net.6.weight": "{}.7.weight", "net.6.bias": "{}.7.bias"}
    for bid in range(1, 6):
        for dst, src in (("conv1.weight", "net.0.weight"), ("conv1.bias", "net.0.bias"), ("conv1.relu", "net.1.weight"),
                         ("conv2.weight", "net.2.weight"), ("conv2.bias", "net.2.bias"), ("conv2.relu", "net.3.weight")):
            m["net.{}.{}".format(bid, dst)] = "{}.%d.%s" % (bid + 1, src)
    return {k: v.format(prefix) for k, v in m.items()}


def cast_entropy_parameter(pdict, ndict):
    """Fill the [3,...] batched parameters from ent.{weight,delta,mean}_net.* (lic360_demo.py:296-311)."""
    for idx, prex in enumerate(["ent.weight_net", "ent.delta_net", "ent.mean_net"]):
        rd = _key_map(prex)
        for pk in ndict.keys():
            ndict[pk][idx] = pdict[rd[pk]]
    return ndict


def cast_imp_entropy_parameter(pdict, ndict):
    rd = _key_map("imp_ent.net")
    for pk in ndict.keys():
        ndict[pk] = pdict[rd[pk]]
    return ndict
