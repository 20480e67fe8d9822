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
    """Shared plumbing of the four plane-sweep drivers.  A driver owns `ctx` (scan order), `net` (the context model), the plane
    gather ops and `mcoder`; its forward() is one sweep over the P = h + w + ngroup - 2 anti-diagonal planes.  Constructor
    signatures and sub-module names are the reference's (they fix the state-dict layout, test/lic360_demo.py:95-290); the sweeps
    are written against the helpers below."""
    cuda_name = "cuda:0"

    def _begin(self, ref_tensor):
        self.p1, self.p2 = self.ctx(ref_tensor)
        self.apply(lambda m: init_entropy_network(m, self.p1, self.p2))

    def _planes(self, h, w):
        return h + w + self.ngroup - 2

    def _open(self, code_name, decode):
        """stateful plane ops back to plane 0, coder onto `code_name`"""
        self.apply(restart_entropy_network)
        self.mcoder.reset_fname(code_name)
        (self.mcoder.start_decoder if decode else self.mcoder.start_encoder)()

    @staticmethod
    def _host_i32(t, cols=None):
        t = t.to(torch.int32).cpu()
        return t.view(-1, cols) if cols else t.view(-1)

    @staticmethod
    def _host_f32(t):
        return t.to(torch.float32).cpu().contiguous().view(-1)

    def _plane_tables(self, params, table_op, ncols):
        """current plane of the context model's output -> (host int32 CDF rows, symbol count); advances `self.ext`"""
        rows, count = self.ext(params)
        return self._host_i32(table_op(rows, count), ncols), int(count[0].item())

    def _device_plane(self, host_symbols, h, w):
        return host_symbols.to(self.cuda_name).view(1, 1, h, w).contiguous()


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
        self._open(code_name, decode=False)

    @torch.no_grad()
    def forward(self, data, mask):
        """symbols `data` in {0..bin_num-1} where mask = 1 -> bitstream.  The model sees the centred, masked symbols once per
        stacked net (weight / sigma / mean); every plane then costs one gather of its 3 x 3 mixture parameters, one table build and
        one pass of the coder over the plane's symbols."""
        self._begin(data)
        centred = ((data - self.bias) * mask).contiguous()
        params = self.net(centred.repeat(3, 1, 1, 1))
        for _plane in range(self._planes(*data.shape[2:])):
            tables, count = self._plane_tables(params, self.gmm, self.bin_num + 1)
            symbols = self._host_i32(self.ext_label(data)[0])
            coded = self._host_f32(self.ext_mask(mask)[0])
            self.mcoder.encodes_mask(tables, self.bin_num, symbols, coded, count)
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
        self._open(code_name, decode=False)

    @torch.no_grad()
    def forward(self, data):
        """importance levels in {0..bin_num} -> bitstream: one group, softmax tables of nsym entries, nothing masked"""
        levels = data.contiguous()
        self._begin(levels)
        logits = self.net(self.scale(levels))
        for _plane in range(self._planes(*levels.shape[2:])):
            tables, count = self._plane_tables(logits, self.table, self.nsym + 1)
            self.mcoder.encodes(tables, self.nsym, self._host_i32(self.ext_label(levels)[0]), count)
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
        self._open(code_name, decode=True)

    @torch.no_grad()
    def forward(self, mask):
        """bitstream + mask -> symbols.  Plane p of the model can only be evaluated once planes < p are decoded: `ipt` scatters the
        previous plane's symbols into the (persistent) model input, the decode-order convolutions extend their outputs by plane p,
        and the coder turns the plane's tables into symbols (3.5 = "not coded" at masked positions)."""
        h, w = mask.shape[2:]
        decoded = torch.zeros((1, 1, h, w), dtype=torch.float32, device=self.cuda_name)
        self._begin(decoded)
        for _plane in range(self._planes(h, w)):
            tables, count = self._plane_tables(self.net(self.ipt(decoded)), self.gmm, self.bin_num + 1)
            coded = self.ext_mask(mask)[0].cpu().contiguous().view(-1)
            decoded = self._device_plane(self.mcoder.decodes_mask(tables, self.bin_num, coded, count), h, w)
        centred = self.ipt(decoded)[0:1]                                  # the last plane's symbols go in with one more scatter
        return (centred + self.bias * mask).contiguous()


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
        self._open(code_name, decode=True)

    @torch.no_grad()
    def forward(self, h=32, w=64):
        """bitstream -> importance levels [1,1,h,w] (kept in `last_levels`) -> the latent's 0/1 mask [1, 4*bin_num/4, 2h, 2w]"""
        decoded = torch.zeros((1, 1, h, w), dtype=torch.float32, device=self.cuda_name)
        self._begin(decoded)
        for _plane in range(self._planes(h, w)):
            tables, count = self._plane_tables(self.net(self.ipt(decoded)), self.table, self.nsym + 1)
            decoded = self._device_plane(self.mcoder.decodes(tables, self.nsym, count), h, w)
        model_input = self.ipt(decoded)                                   # = level * scale - 1
        self.last_levels = torch.floor((model_input + 1) / self.scale + 1e-5).to(torch.float32).contiguous()
        return self.d2w(self.i2m(self.last_levels))


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
