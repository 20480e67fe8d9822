"""TEST INFRASTRUCTURE ONLY -- a CPU stand-in for the reference's compiled extension module `lic360`
(extension/main.cpp:4-178), just complete enough for the reference's OWN Python (lic360_operator/*.py wrappers and the codec
drivers of test/lic360_demo.py:95-322) to run in the build container, where neither nvcc nor a GPU exists.

Every kernel call goes to the CPU oracle (oracle/liblic360_oracle.so through oracle/oracle.py: the restatement of the reference's
CUDA kernels); the arithmetic coder is the oracle's, and -- when oracle/_ref/libref_ac.so is built -- every finished bitstream is
re-encoded with the REFERENCE's ArithmeticCoder.cpp and must come out byte-identical.  What this module adds is only the per-op
state the extension's C++ classes keep (plan_sum_ counters, op-owned output buffers, set_param / restart), restated from
extension/*.hpp / *_cuda.cu (cited per class).  oracle/gen_golden_drivers.py imports the reference's drivers over it and writes
tests/golden/driver_*.npz; nothing in the product or on the GPU box imports this file.

The 12 classes the four codec drivers use are implemented; the other 14 bound names exist (the reference's wrappers construct
them by name) and say so when called.
"""
import numpy as np
import torch

import oracle as orc


def _np(t, dtype=np.float32):
    a = t.detach().contiguous().numpy()
    assert a.dtype == dtype, (a.dtype, dtype)
    return a


class _Op(object):
    """base_opt (extension/base_opt.hpp:4-81): device, shape cache, op-owned outputs"""

    def __init__(self, device=0, timeit=False):
        self.device_ = device
        self._shape = None
        self._top = None

    def to(self, device):                            # BaseOpModule.custom_op_to (lic360_operator/BaseOpModule.py:33-39)
        self.device_ = device

    def _reshape(self, shape):
        shape = tuple(int(s) for s in shape)
        if shape == self._shape:
            return False
        self._shape = shape
        return True

    def _buf(self, shape):
        if self._top is None or tuple(self._top.shape) != tuple(shape):
            self._top = torch.zeros(shape, dtype=torch.float32)
        return self._top


class CodeContexOp(_Op):
    """extension/code_contex_cuda.cu:11-38: scan order idx[2][HW] + plane prefix (CPU int32)"""

    def forward(self, x):
        h, w = int(x.shape[2]), int(x.shape[3])
        idx, pidx = orc.code_contex(h, w)
        return [torch.from_numpy(idx).view(h, w, 2), torch.from_numpy(pidx)]


class _PlaneOp(_Op):
    """plan_sum_ / set_param / restart (extension/cconv_dc.hpp:21-27 and its twins in tile_*.hpp)"""

    def __init__(self, ngroup, device=0, timeit=False):
        super().__init__(device, timeit)
        self.ngroup_ = int(ngroup)
        self.plan_sum_ = 0
        self.idx_ = self.pidx_ = None

    def restart(self):
        self.plan_sum_ = 0

    def set_param(self, idx, pidx):
        assert idx.dtype == torch.int32 and pidx.dtype == torch.int32
        self.idx_ = np.ascontiguousarray(idx.numpy().reshape(-1))
        self.pidx_ = np.ascontiguousarray(pidx.numpy().reshape(-1))

    def _step(self, shape):
        if self._reshape(shape):
            self.plan_sum_ = 0
        assert self.idx_ is not None, "Slice Index has not been initialized"
        p = self.plan_sum_
        self.plan_sum_ += 1
        return p


class TileExtractOp(_PlaneOp):
    """extension/tile_extract_cuda.cu:48-98 (forward), :120-151 (forward_batch); count returned as CPU int32[1]"""

    def __init__(self, ngroup, label, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)
        self.label_ = bool(label)
        self.top_num_ = torch.zeros((1,), dtype=torch.int32)

    def forward(self, x):
        n, c, h, w = x.shape
        p = self._step(x.shape)
        top = self._buf((n, c // self.ngroup_, h, w))
        self.top_num_[0] = orc.tile_extract(_np(x), top.numpy().reshape(-1), self.ngroup_, self.label_, self.idx_, self.pidx_, p)
        return [top, self.top_num_]

    def forward_batch(self, x):
        n, c, h, w = x.shape
        p = self._step(x.shape)
        top = self._buf((n, c // self.ngroup_, h, w))
        self.top_num_[0] = orc.tile_extract_batch(_np(x), top.numpy().reshape(-1), self.ngroup_, self.idx_, self.pidx_, p)
        return [top, self.top_num_]


class TileInputOp(_PlaneOp):
    """extension/tile_input_cuda.cu:46-76: scatter plane psum-1 as scale*sym+bias, replicated"""

    def __init__(self, ngroup, bias, scale, replicate=1, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)
        self.bias_, self.scale_, self.rep_ = float(bias), float(scale), int(replicate)

    def forward(self, x):
        n, h, w = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
        p = self._step((n, self.ngroup_, h, w))
        top = self._buf((self.rep_ * n, self.ngroup_, h, w))
        orc.tile_input(_np(x).reshape(-1), top.numpy().reshape(-1), n, self.ngroup_, h, w, self.bias_, self.scale_, self.rep_,
                       self.idx_, self.pidx_, p)
        return [top]


class TileAddOp(_PlaneOp):
    """extension/tile_add_cuda.cu:40-60: y += x on the current plane, in place"""

    def forward(self, y, x):
        p = self._step(y.shape)
        assert y.is_contiguous()
        orc.tile_add(y.numpy(), _np(x), self.ngroup_, self.idx_, self.pidx_, p)
        return [y]


class _Conv(_PlaneOp):
    def __init__(self, channel, ngroup, nout, kernel_size, constrain, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)
        self.channel_, self.nout_, self.k_, self.constrain_ = int(channel), int(nout), int(kernel_size), int(constrain)

    def forward(self, x, weight, bias):
        return self._run(x, weight, bias, None)

    def forward_act(self, x, weight, bias, act):
        return self._run(x, weight, bias, act)

    def forward_batch(self, x, weight, bias):
        return self._run(x, weight, bias, None)

    def forward_act_batch(self, x, weight, bias, act):
        return self._run(x, weight, bias, act)


class CconvEcOp(_Conv):
    """extension/cconv_ec_cuda.cu:99-339: whole-tensor masked conv (the batch forms differ by the weight's leading dimension)"""

    def _run(self, x, weight, bias, act):
        out = orc.cconv_ec(_np(x), _np(weight), _np(bias), None if act is None else _np(act), self.ngroup_, self.constrain_)
        return [torch.from_numpy(out)]


class CconvDcOp(_Conv):
    """extension/cconv_dc_cuda.cu:108-398: plane psum only, into the persistent output (memset at psum == 0)"""

    def _run(self, x, weight, bias, act):
        n, c, h, w = x.shape
        p = self._step(x.shape)
        top = self._buf((n, self.nout_, h, w))
        orc.cconv_dc_plane(_np(x), _np(weight), _np(bias), None if act is None else _np(act), top.numpy(), self.ngroup_, self.constrain_,
                           self.idx_, self.pidx_, p)
        return [top]


class EntropyGmmTableOp(_Op):
    """extension/entropy_gmm_table_cuda.cu:109-135 (forward), :161-191 (forward_batch); the count is read on the host (:165)"""

    def __init__(self, nstep, bias, num_gaussian, total_region, beta=1e-6, device=0, timeit=False):
        super().__init__(device, timeit)
        self.nstep_, self.bias_, self.ng_, self.total_, self.beta_ = int(nstep), float(bias), int(num_gaussian), float(total_region), float(beta)

    def forward(self, weight, delta, mean, tnum):
        n, c, h, w = weight.shape
        top = self._buf((n * h * w, self.nstep_ + 1))
        tn = int(tnum[0])
        if tn > 0:
            wv, dv, mv = (_np(t).reshape(-1, self.ng_) for t in (weight, delta, mean))
            top[:tn] = torch.from_numpy(orc.gmm_table(wv, dv, mv, tn, self.nstep_, self.bias_, self.total_, self.beta_))
        return [top]

    def forward_batch(self, data, tnum):
        n, c, h, w = data.shape
        top = self._buf((n * h * w // 3, self.nstep_ + 1))
        tn = int(tnum[0])
        if tn > 0:
            assert data.is_contiguous()
            top[:tn] = torch.from_numpy(orc.gmm_table_batch(data.numpy().reshape(-1), n * c * h * w // 3, tn, self.ng_, self.nstep_,
                                                            self.bias_, self.total_, self.beta_))
        return [top]


class EntropyTableOp(_Op):
    """extension/entropy_table_cuda.cu:78-96"""

    def __init__(self, nstep, totoal_region, device=0, timeit=False):
        super().__init__(device, timeit)
        self.nstep_, self.total_ = int(nstep), float(totoal_region)

    def forward(self, data, count_tensor):
        n, c, h, w = data.shape
        top = self._buf((n * h * w, self.nstep_ + 1))
        cnt = int(count_tensor[0])
        if cnt > 0:
            top[:cnt] = torch.from_numpy(orc.entropy_table(_np(data).reshape(-1), cnt, self.nstep_, self.total_))
        return [top]


class ScaleOp(_Op):
    """extension/scale_cuda.cu:32-48"""

    def __init__(self, bias, scale, device=0, timeit=False):
        super().__init__(device, timeit)
        self.bias_, self.scale_ = float(bias), float(scale)

    def forward(self, x):
        return [torch.from_numpy(orc.scale(_np(x), self.bias_, self.scale_))]


class Imp2maskOp(_Op):
    """extension/imp2mask_cuda.cu:41-57"""

    def __init__(self, levels, channels, device=0, timeit=False):
        super().__init__(device, timeit)
        self.levels_, self.channels_ = int(levels), int(channels)

    def forward(self, x):
        return [torch.from_numpy(orc.imp2mask(_np(x), self.levels_, self.channels_))]


class DtowOp(_Op):
    """extension/dtow_cuda.cu:77-102"""

    def __init__(self, stride=2, d2w=True, device=0, timeit=False):
        super().__init__(device, timeit)
        self.stride_, self.d2w_ = int(stride), bool(d2w)

    def forward(self, x):
        return [torch.from_numpy(orc.dtow(_np(x), self.stride_, self.d2w_))]


class Coder(object):
    """extension/coder.h:10-63, coder.cpp:30-113: per-plane slices into one arithmetic-coded file"""

    def __init__(self, name, file_value):
        self.fname, self.file_value = str(name), float(file_value)
        self._enc = self._dec = None
        self._log = []                               # every encoded slice, for the cross-check with the reference coder

    def reset_fname(self, name):
        self.fname = str(name)

    def get_fname(self):
        return self.fname

    def start_encoder(self):
        self._enc, self._log = orc.Encoder(), []

    def _encode(self, table, ncode, label, mask, num):
        t = np.ascontiguousarray(table.numpy().reshape(-1, ncode + 1)[:num], np.int32)
        lab = np.ascontiguousarray(label.numpy().reshape(-1)[:num], np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask.numpy().reshape(-1)[:num], np.float32)
        self._enc.encode(t, ncode, lab, mk, num)
        self._log.append((ncode, t.copy(), lab.copy(), None if mk is None else mk.copy()))

    def encodes(self, table, ncode, label, num):
        self._encode(table, int(ncode), label, None, int(num))

    def encodes_mask(self, table, ncode, label, mask, num):
        self._encode(table, int(ncode), label, mask, int(num))

    def end_encoder(self):
        data = self._enc.finish()
        self._enc = None
        if orc.have_ref() and self._log:             # the same symbols through the REFERENCE's ArithmeticCoder.cpp, in one go
            ncode = self._log[0][0]
            tabs = np.concatenate([s[1] for s in self._log])
            labs = np.concatenate([s[2] for s in self._log])
            masked = any(s[3] is not None for s in self._log)
            mks = np.concatenate([s[3] if s[3] is not None else np.ones(len(s[2]), np.float32) for s in self._log]) if masked else None
            ref = orc.ref_encode(tabs, ncode, labs, mks) if len(labs) else data
            assert ref == data, "oracle coder and reference coder disagree on a driver's bitstream"
        with open(self.fname, "wb") as f:
            f.write(data)

    def start_decoder(self):
        with open(self.fname, "rb") as f:
            self._dec = orc.Decoder(f.read())

    def decodes(self, table, ncode, num):
        rows = int(table.shape[0])
        out = self._dec.decode(table.numpy().reshape(rows, -1), int(ncode), None, int(num), self.file_value, size=rows)
        return torch.from_numpy(out)

    def decodes_mask(self, table, ncode, mask, num):
        rows = int(table.shape[0])
        mk = np.ascontiguousarray(mask.numpy().reshape(-1), np.float32)
        out = self._dec.decode(table.numpy().reshape(rows, -1), int(ncode), mk, int(num), self.file_value, size=rows)
        return torch.from_numpy(out)


def _unused(name, where):
    class _Placeholder(_Op):
        def __init__(self, *a, **k):
            super().__init__()

        def __getattr__(self, attr):
            raise NotImplementedError("lic360.%s (%s) is not on the codec drivers' path; this CPU stand-in only carries test/lic360_demo.py:95-322" % (name, where))
    _Placeholder.__name__ = name
    return _Placeholder


for _n, _w in (("ProjectsOp", "projects_cuda.cu"), ("SpherePadOp", "sphere_pad_cuda.cu"), ("SphereTrimOp", "sphere_trim_cuda.cu"),
               ("SphereCutEdgeOp", "sphere_cut_edge_cuda.cu"), ("ImpMapOp", "imp_map_cuda.cu"), ("QuantOp", "quant_cuda.cu"),
               ("SphereLatScaleOp", "sphere_lat_scale_cuda.cu"), ("ContexShiftOp", "contex_shift_cuda.cu"),
               ("ContextReshapeOp", "context_reshape_cuda.cu"), ("EntropyGmmOp", "entropy_gmm_cuda.cu"),
               ("MaskConstrainOp", "mask_constrain_cuda.cu"), ("DquantOp", "dquant_cuda.cu"), ("CppOp", "CPP_cuda.cu"),
               ("ViewportOp", "viewport_cuda.cu")):
    globals()[_n] = _unused(_n, _w)
