"""`lic360` -- drop-in replacement of the reference's pybind11 extension module
(extension/main.cpp:4-178) on top of liblic360_hip.so (hand-written HIP for gfx950).

Same class names, constructor arities and method names as the reference's bound classes, so that
`lic360_operator/*` and `test/lic360_demo.py` style drivers run unchanged on PyTorch-ROCm.  torch is
used here only as plumbing (device memory, streams); every computation goes through the C ABI
declared in include/lic360_hip.h.  There is NO CPU fallback: without the shared library or without
a HIP device the ops raise.

Reference semantics kept on purpose:
  * outputs are op-owned buffers re-used across calls (extension/base_opt.hpp:43-57);
  * plane-stepped ops keep a `plan_sum_` counter, `restart()` and `set_param(idx_gpu, plane_idx_cpu)`
    (extension/cconv_dc.hpp:21-27); shape changes reset the counter;
  * EntropyGmmTable rewrites its inputs in place; SpherePad(inplace)/SphereTrim/TileAdd mutate inputs;
  * Coder works on CPU int32 tensors and files.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "liblic360_hip.so")


class Lic360Error(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "liblic360_hip.so not found at %s -- build it with `python __graft_entry__.py` "
            "(or `make -C 360-image-compression_amd/csrc`); there is no CPU fallback" % _LIB_PATH)
    return C.CDLL(_LIB_PATH)


_lib = _load()
_lib.lic360_last_error.restype = C.c_char_p
_lib.lic360_conv_plan_packed_floats.restype = C.c_long
_lib.lic360_conv_plan_packed_floats.argtypes = [C.c_void_p]
_lib.lic360_conv_plan_destroy.argtypes = [C.c_void_p]
_lib.lic360_coder_enc_open.restype = C.c_void_p
_lib.lic360_coder_dec_open.restype = C.c_void_p
_lib.lic360_coder_enc_finish.restype = C.c_long
_lib.lic360_coder_bytes.restype = C.POINTER(C.c_uint8)

# exported for tests (symbol presence check against include/lic360_hip.h)
LIBRARY_PATH = _LIB_PATH


def _chk(rc):
    if rc != 0:
        raise Lic360Error(_lib.lic360_last_error().decode())


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f(v):
    return C.c_float(float(v))


def _check_in(t, device, name="input"):
    if not t.is_cuda:
        raise Lic360Error("%s must be a CUDA(HIP) tensor on device %d" % (name, device))
    if t.dtype != torch.float32:
        raise Lic360Error("%s must be float32" % name)
    if not t.is_contiguous():
        raise Lic360Error("%s must be contiguous" % name)


class _Op(object):
    """Counterpart of base_opt (extension/base_opt.hpp:4-81)."""

    def __init__(self, device=0, timeit=False):
        self.device_ = int(device)
        self.timeit_ = bool(timeit)
        self._shape = None
        self._top = []

    def to(self, device):
        device = int(device)
        if device != self.device_:
            self.device_ = device
            self._shape = None
            self._top = []
            self._moved()

    def _moved(self):
        pass

    def _reshape(self, shape):
        """True when the input shape changed (reshape_base)."""
        shape = tuple(int(s) for s in shape)
        if shape == self._shape:
            return False
        self._shape = shape
        return True

    def _tops(self, like, shapes):
        """(Re)allocate op-owned outputs when the first shape differs (reshape_top_base)."""
        if not self._top or tuple(self._top[0].shape) != tuple(shapes[0]) or self._top[0].device != like.device:
            self._top = [torch.empty(s, dtype=torch.float32, device=like.device) for s in shapes]
            return True
        return False

    def _s(self):
        return _stream(self.device_)


# --------------------------------------------------------------------------------------- sphere ops
class SpherePadOp(_Op):
    def __init__(self, pad, inplace=False, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_, self.inplace_ = int(pad), bool(inplace)

    def forward(self, x):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        self._reshape(x.shape)
        if self.inplace_:
            _chk(_lib.lic360_sphere_pad_inplace(self._s(), _p(x), n * c, h, w, self.pad_))
            return [x]
        self._tops(x, [(n, c, h + 2 * self.pad_, w + 2 * self.pad_)])
        _chk(_lib.lic360_sphere_pad(self._s(), _p(x), _p(self._top[0]), n * c, h, w, self.pad_))
        return self._top

    def backward(self, top_diff):
        raise NotImplementedError("SpherePadOp.backward is training-side (SURVEY.md §8f.4): out of scope")


class SphereTrimOp(_Op):
    def __init__(self, pad=1, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_ = int(pad)

    def forward(self, x):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        _chk(_lib.lic360_sphere_trim(self._s(), _p(x), n * c, h, w, self.pad_))
        return [x]

    backward = forward          # extension/sphere_trim_cuda.cu:48-66: same kernel on the gradient


class SphereCutEdgeOp(_Op):
    def __init__(self, pad=1, device=0, timeit=False):
        super().__init__(device, timeit)
        self.pad_ = int(pad)

    def forward(self, x):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        self._tops(x, [(n, c, h - 2 * self.pad_, w - 2 * self.pad_)])
        _chk(_lib.lic360_sphere_cut_edge(self._s(), _p(x), _p(self._top[0]), n * c, h, w, self.pad_))
        return self._top

    def backward(self, top_diff):
        raise NotImplementedError("SphereCutEdgeOp.backward is training-side: out of scope")


class SphereLatScaleOp(_Op):
    def __init__(self, npart, device=0, timeit=False):
        super().__init__(device, timeit)
        self.npart_ = int(npart)

    def set_npart(self, npart):
        self.npart_ = int(npart)
        self._shape = None

    def _run(self, x, weight):
        _check_in(x, self.device_)
        _check_in(weight, self.device_, "weight")
        n, c, h, w = x.shape
        if h % self.npart_:
            raise Lic360Error("height must be a multiple of npart")
        out = torch.empty_like(x)
        _chk(_lib.lic360_sphere_lat_scale(self._s(), _p(x), _p(weight), _p(out), n * c, h, w, self.npart_))
        return out

    def forward(self, x, weight):
        self._top = [self._run(x, weight)]
        return self._top

    def backward(self, top_diff, weight):           # extension/sphere_lat_scale_cuda.cu:69-87
        return [self._run(top_diff, weight)]


# --------------------------------------------------------------------------------------- pointwise ops
class ImpMapOp(_Op):
    def __init__(self, levels, alpha, gamma, rt, scale_constrain, scale_weight, imp_kernel=0, ntop=1, device=0, timeit=False):
        super().__init__(device, timeit)
        self.levels_, self.alpha_, self.gamma_, self.rt_ = int(levels), float(alpha), float(gamma), float(rt)
        self.scale_constrain_, self.scale_weight_ = float(scale_constrain), float(scale_weight)
        self.imp_kernel_, self.ntop_ = int(imp_kernel), int(ntop)

    def forward(self, x, imp):
        _check_in(x, self.device_)
        _check_in(imp, self.device_, "imp")
        n, c, h, w = x.shape
        if c % self.levels_:
            raise Lic360Error("channels must be a multiple of levels")
        shapes = [(n, c, h, w), (n, 1, h)] + ([(n, c, h, w)] if self.ntop_ > 1 else [])
        if self._tops(x, shapes):
            _chk(_lib.lic360_imp_map_constrain(self._s(), _p(self._top[1]), n, h, _f(self.rt_), _f(self.scale_constrain_)))
        mask = self._top[2] if self.ntop_ > 1 else None
        _chk(_lib.lic360_imp_map(self._s(), _p(x), _p(imp), _p(self._top[0]), _p(mask), n, c, h, w, self.levels_))
        return self._top

    def backward(self, *a):
        raise NotImplementedError("ImpMapOp.backward is training-side: out of scope")


class Imp2maskOp(_Op):
    def __init__(self, levels, channels, device=0, timeit=False):
        super().__init__(device, timeit)
        self.levels_, self.channel_ = int(levels), int(channels)

    def forward(self, x):
        _check_in(x, self.device_)
        n, _, h, w = x.shape
        self._tops(x, [(n, self.channel_, h, w)])
        _chk(_lib.lic360_imp2mask(self._s(), _p(x), _p(self._top[0]), n, self.channel_, h, w, self.channel_ // self.levels_))
        return self._top


class ScaleOp(_Op):
    def __init__(self, bias, scale, device=0, timeit=False):
        super().__init__(device, timeit)
        self.bias_, self.scale_ = float(bias), float(scale)

    def forward(self, x):
        _check_in(x, self.device_)
        self._tops(x, [tuple(x.shape)])
        _chk(_lib.lic360_scale(self._s(), _p(x), _p(self._top[0]), C.c_long(x.numel()), _f(self.bias_), _f(self.scale_)))
        return self._top


class QuantOp(_Op):
    def __init__(self, channel, bin_num, weight_decay=0.9, check_iters=100, ntop=1, top_alpha=0.1, device=0, timeit=False):
        super().__init__(device, timeit)
        self.channel_, self.bin_num_, self.ntop_ = int(channel), int(bin_num), int(ntop)
        self.weight_ = None
        self.count_data_ = None

    def forward(self, x, weight, ncount, train):
        if train:
            raise NotImplementedError("QuantOp training path (update_weight) is out of scope")
        _check_in(x, self.device_)
        _check_in(weight, self.device_, "weight")
        n, c, h, w = x.shape
        if c != self.channel_ or tuple(weight.shape) != (self.channel_, self.bin_num_):
            raise Lic360Error("QuantOp shape mismatch")
        if self.weight_ is None or self.weight_.device != x.device:
            self.weight_ = torch.zeros((c, self.bin_num_), dtype=torch.float32, device=x.device)
            self.count_data_ = torch.zeros((c, self.bin_num_), dtype=torch.float32, device=x.device)
        self._tops(x, [tuple(x.shape)] * (2 if self.ntop_ > 1 else 1))
        qidx = self._top[1] if self.ntop_ > 1 else None
        _chk(_lib.lic360_quant(self._s(), _p(x), _p(weight), _p(self.weight_), _p(self._top[0]), _p(qidx), _p(self.count_data_),
                               n, c, h, w, self.bin_num_))
        return self._top

    def backward(self, *a):
        raise NotImplementedError("QuantOp.backward is training-side: out of scope")


class DquantOp(_Op):
    def __init__(self, channel, bin_num, device=0, timeit=False):
        super().__init__(device, timeit)
        self.nchannel_, self.bin_num_ = int(channel), int(bin_num)
        self.weight_ = None

    def forward(self, x, mask, weight):
        for t, nm in ((x, "input"), (mask, "mask"), (weight, "weight")):
            _check_in(t, self.device_, nm)
        n, c, h, w = x.shape
        if self.weight_ is None or self.weight_.device != x.device:
            self.weight_ = torch.zeros((self.nchannel_, self.bin_num_), dtype=torch.float32, device=x.device)
        self._tops(x, [tuple(x.shape)])
        _chk(_lib.lic360_dquant(self._s(), _p(x), _p(mask), _p(weight), _p(self.weight_), _p(self._top[0]), n, c, h, w, self.bin_num_))
        return self._top


class DtowOp(_Op):
    def __init__(self, stride=2, d2w=True, device=0, timeit=False):
        super().__init__(device, timeit)
        self.stride_, self.d2w_ = int(stride), bool(d2w)

    def _run(self, x, d2w):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        s = self.stride_
        shape = (n, c // (s * s), h * s, w * s) if d2w else (n, c * s * s, h // s, w // s)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        _chk(_lib.lic360_dtow(self._s(), _p(x), _p(out), n, c, h, w, s, int(d2w)))
        return out

    def forward(self, x):
        n, c, h, w = x.shape
        s = self.stride_
        shape = (n, c // (s * s), h * s, w * s) if self.d2w_ else (n, c * s * s, h // s, w // s)
        _check_in(x, self.device_)
        self._tops(x, [shape])
        _chk(_lib.lic360_dtow(self._s(), _p(x), _p(self._top[0]), n, c, h, w, s, int(self.d2w_)))
        return self._top

    def backward(self, top_diff):                    # the inverse permutation (extension/dtow_cuda.cu:104-160)
        return [self._run(top_diff, not self.d2w_)]


# --------------------------------------------------------------------------------------- layout ops
class ContextReshapeOp(_Op):
    def __init__(self, ngroup, device=0, timeit=False):
        super().__init__(device, timeit)
        self.ngroup_ = int(ngroup)

    def forward(self, x):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        self._reshape(x.shape)
        self._tops(x, [(n * h * w * self.ngroup_, c // self.ngroup_)])
        _chk(_lib.lic360_context_reshape(self._s(), _p(x), _p(self._top[0]), n, c, h, w, self.ngroup_, 0))
        return self._top

    def backward(self, top_diff):
        _check_in(top_diff, self.device_)
        n, c, h, w = self._shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=top_diff.device)
        _chk(_lib.lic360_context_reshape(self._s(), _p(top_diff), _p(out), n, c, h, w, self.ngroup_, 1))
        return [out]


class ContexShiftOp(_Op):
    def __init__(self, inv, cpn=1, device=0, timeit=False):
        super().__init__(device, timeit)
        self.inv_, self.cpn_ = bool(inv), int(cpn)

    def forward(self, x):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        g = c // self.cpn_
        ho = h - w - g + 2 if self.inv_ else h + w + g - 2
        self._tops(x, [(n, c, ho, w)])
        _chk(_lib.lic360_contex_shift(self._s(), _p(x), _p(self._top[0]), n, c, h, w, self.cpn_, int(self.inv_)))
        return self._top

    def backward(self, top_diff):
        _check_in(top_diff, self.device_)
        n, c, h, w = top_diff.shape
        g = c // self.cpn_
        ho = h + w + g - 2 if self.inv_ else h - w - g + 2
        out = torch.empty((n, c, ho, w), dtype=torch.float32, device=top_diff.device)
        _chk(_lib.lic360_contex_shift(self._s(), _p(top_diff), _p(out), n, c, h, w, self.cpn_, int(not self.inv_)))
        return [out]


# --------------------------------------------------------------------------------------- scan order + plane ops
class CodeContexOp(_Op):
    def __init__(self, device=0, timeit=False):
        super().__init__(device, timeit)
        self.idx_mat_ = None
        self.plane_idx_ = None

    def forward(self, x):
        h, w = int(x.shape[2]), int(x.shape[3])
        if self._reshape(x.shape) or self.idx_mat_ is None:
            idx = torch.zeros((h, w, 2), dtype=torch.int32)
            pidx = torch.zeros((h + w,), dtype=torch.int32)
            _chk(_lib.lic360_code_contex(h, w, _p(idx), _p(pidx)))
            self.idx_mat_ = idx.to("cuda:%d" % self.device_)
            self.plane_idx_ = pidx
        return [self.idx_mat_, self.plane_idx_]

    def backward(self, top_diff):
        return []


class _PlaneOp(_Op):
    """plan_sum_/set_param/restart state shared by the plane-stepped ops."""

    def __init__(self, ngroup, device, timeit):
        super().__init__(device, timeit)
        self.ngroup_ = int(ngroup)
        self.plan_sum_ = 0
        self.param_set_ = False
        self.index_mat_ = None
        self.plan_idx_mat_ = None
        self.plan_idx_dev_ = None

    def _moved(self):
        self.param_set_ = False

    def restart(self):
        self.plan_sum_ = 0

    def set_param(self, idx, pidx):
        if not idx.is_cuda or idx.dtype != torch.int32:
            raise Lic360Error("set_param: idx must be an int32 device tensor")
        if pidx.is_cuda or pidx.dtype != torch.int32:
            raise Lic360Error("set_param: plane_idx must be an int32 CPU tensor (it is read on the host)")
        self.index_mat_ = idx.contiguous()
        self.plan_idx_mat_ = pidx.contiguous()
        self.plan_idx_dev_ = None
        self.param_set_ = True

    def _plane_shape(self, shape):
        if self._reshape(shape):
            self.plan_sum_ = 0
            if not self.param_set_:
                raise Lic360Error("Slice Index has not been initialized (call set_param first)")

    def _window(self, psum, h, w):
        start, ln = C.c_int(0), C.c_int(0)
        _chk(_lib.lic360_plane_window(psum, self.ngroup_, h, w, _p(self.plan_idx_mat_), C.byref(start), C.byref(ln)))
        return start.value, ln.value

    def _pidx_dev(self):
        if self.plan_idx_dev_ is None:
            self.plan_idx_dev_ = self.plan_idx_mat_.to(self.index_mat_.device)
        return self.plan_idx_dev_


class TileExtractOp(_PlaneOp):
    def __init__(self, ngroup, label, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)
        self.label_ = bool(label)
        self.top_num_ = torch.zeros((1,), dtype=torch.int32)

    def _prep(self, x):
        _check_in(x, self.device_)
        n, c, h, w = x.shape
        if self._reshape(x.shape):
            self.plan_sum_ = 0
            self.top_num_ = torch.zeros((1,), dtype=torch.int32)
            if not self.param_set_:
                raise Lic360Error("Slice Index has not been initialized (call set_param first)")
        cpn = c // self.ngroup_
        self._tops(x, [(n, cpn, h, w)])
        return n, c, h, w, cpn

    def forward(self, x):
        n, c, h, w, cpn = self._prep(x)
        psum = self.plan_sum_
        self.plan_sum_ += 1
        mod = h + w + self.ngroup_ - 2
        self.top_num_[0] = 0
        top = self._top[0]
        if self.label_:
            if psum < mod:
                start, ln = self._window(psum, h, w)
                self.top_num_[0] = n * ln
                if ln > 0:
                    _chk(_lib.lic360_tile_extract(self._s(), _p(x), _p(top), n, c, h, w, self.ngroup_, _p(self.index_mat_), start, ln, psum))
        else:
            if psum == 0:
                top.zero_()
            elif psum <= mod:
                psum -= 1
                start, ln = self._window(psum, h, w)
                self.top_num_[0] = n * ln
                if ln > 0:
                    _chk(_lib.lic360_tile_extract(self._s(), _p(x), _p(top), n, c, h, w, self.ngroup_, _p(self.index_mat_), start, ln, psum))
        return [top, self.top_num_]

    def forward_batch(self, x):
        n, c, h, w, cpn = self._prep(x)
        psum = self.plan_sum_
        self.plan_sum_ += 1
        self.top_num_[0] = 0
        if psum < h + w + self.ngroup_ - 2:
            start, ln = self._window(psum, h, w)
            self.top_num_[0] = (n // 3) * ln
            if ln > 0:
                _chk(_lib.lic360_tile_extract_batch(self._s(), _p(x), _p(self._top[0]), n, c, h, w, self.ngroup_, _p(self.index_mat_), start, ln, psum))
        return [self._top[0], self.top_num_]


class TileInputOp(_PlaneOp):
    def __init__(self, ngroup, bias, scale, replicate=1, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)
        self.bias_, self.scale_, self.rep_ = float(bias), float(scale), int(replicate)

    def forward(self, x):
        _check_in(x, self.device_)
        n, h, w = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
        self._plane_shape((n, self.ngroup_, h, w))
        self._tops(x, [(self.rep_ * n, self.ngroup_, h, w)])
        psum = self.plan_sum_
        self.plan_sum_ += 1
        top = self._top[0]
        if psum == 0:
            top.zero_()
        elif psum <= h + w + self.ngroup_ - 2:
            psum -= 1
            start, ln = self._window(psum, h, w)
            _chk(_lib.lic360_tile_input(self._s(), _p(x), _p(top), n, self.ngroup_, h, w, _f(self.bias_), _f(self.scale_), self.rep_,
                                        _p(self.index_mat_), start, ln, psum))
        return self._top


class TileAddOp(_PlaneOp):
    def __init__(self, ngroup, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)

    def forward(self, y, x):
        _check_in(y, self.device_)
        _check_in(x, self.device_)
        n, c, h, w = y.shape
        self._plane_shape(y.shape)
        psum = self.plan_sum_
        self.plan_sum_ += 1
        start, ln = self._window(psum, h, w)
        if ln > 0:
            _chk(_lib.lic360_tile_add(self._s(), _p(y), _p(x), n, c, h, w, self.ngroup_, _p(self.index_mat_), start, ln, psum))
        return [y]


# --------------------------------------------------------------------------------------- tables
class EntropyGmmTableOp(_Op):
    def __init__(self, nstep, bias, num_gaussian, total_region, beta=1e-6, device=0, timeit=False):
        super().__init__(device, timeit)
        self.nstep_, self.bias_, self.ng_ = int(nstep), float(bias), int(num_gaussian)
        self.total_, self.beta_ = int(total_region), float(beta)

    def forward(self, weight, delta, mean, tnum):
        for t in (weight, delta, mean):
            _check_in(t, self.device_)
        n, c, h, w = weight.shape
        self._tops(weight, [(n * h * w, self.nstep_ + 1)])
        tn = int(tnum[0])
        _chk(_lib.lic360_gmm_table(self._s(), _p(weight), _p(delta), _p(mean), _p(self._top[0]), tn, self.ng_, self.nstep_,
                                   _f(self.bias_), _f(self.total_), _f(self.beta_)))
        return self._top

    def forward_batch(self, data, tnum):
        _check_in(data, self.device_)
        n, c, h, w = data.shape
        self._tops(data, [(n * h * w // 3, self.nstep_ + 1)])
        tn = int(tnum[0])
        stride = n * c * h * w // 3
        if tn > 0:
            base = data.data_ptr()
            _chk(_lib.lic360_gmm_table(self._s(), C.c_void_p(base), C.c_void_p(base + 4 * stride), C.c_void_p(base + 8 * stride),
                                       _p(self._top[0]), tn, self.ng_, self.nstep_, _f(self.bias_), _f(self.total_), _f(self.beta_)))
        return self._top


class EntropyTableOp(_Op):
    def __init__(self, nstep, totoal_region, device=0, timeit=False):
        super().__init__(device, timeit)
        self.nstep_, self.total_ = int(nstep), float(totoal_region)

    def forward(self, data, count_tensor):
        _check_in(data, self.device_)
        n, c, h, w = data.shape
        self._tops(data, [(n * h * w, self.nstep_ + 1)])
        _chk(_lib.lic360_entropy_table(self._s(), _p(data), _p(self._top[0]), int(count_tensor[0]), self.nstep_, _f(self.total_)))
        return self._top


class EntropyGmmOp(_Op):
    def __init__(self, num_gaussian=3, ignore_label=-1, device=0, timeit=False):
        super().__init__(device, timeit)
        self.ng_ = int(num_gaussian)
        self._diff = None

    def forward(self, weight, delta, mean, label):
        for t in (weight, delta, mean, label):
            _check_in(t, self.device_)
        m, ng = weight.shape
        if ng != self.ng_:
            raise Lic360Error("last dim of weight must equal num_gaussian")
        self._tops(weight, [(m,)])
        if self._diff is None or self._diff[0].shape[0] != m or self._diff[0].device != weight.device:
            mk = lambda s: torch.empty(s, dtype=torch.float32, device=weight.device)
            self._diff = [mk((m, ng)), mk((m, ng)), mk((m, ng)), mk((m, 1))]
        d = self._diff
        _chk(_lib.lic360_entropy_gmm(self._s(), _p(weight), _p(delta), _p(mean), _p(label), _p(self._top[0]),
                                     _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), m, ng))
        return self._top

    def backward(self, top_diff):
        d = self._diff
        _check_in(top_diff, self.device_)
        _chk(_lib.lic360_entropy_gmm_backward(self._s(), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(top_diff), d[0].shape[0], self.ng_))
        return d


# --------------------------------------------------------------------------------------- masked convolution
class _ConvBase(_PlaneOp):
    def __init__(self, channel, ngroup, nout, kernel_size, constrain, device=0, timeit=False):
        super().__init__(ngroup, device, timeit)
        self.channel_, self.nout_, self.kernel_size_, self.constrain_ = int(channel), int(nout), int(kernel_size), int(constrain)
        self._plan = None
        self._packed = None
        self._packed_key = None

    def __del__(self):
        try:
            if self._plan is not None:
                _lib.lic360_conv_plan_destroy(self._plan)
        except Exception:
            pass

    def _moved(self):
        self.param_set_ = False
        if self._plan is not None:
            _lib.lic360_conv_plan_destroy(self._plan)
        self._plan, self._packed, self._packed_key = None, None, None

    def _get_plan(self):
        if self._plan is None:
            with torch.cuda.device(self.device_):
                h = C.c_void_p(0)
                _chk(_lib.lic360_conv_plan_create(self.channel_, self.ngroup_, self.nout_, self.kernel_size_, self.constrain_, C.byref(h)))
                self._plan = h
        return self._plan

    def _pack(self, weight, nb):
        """A-fragment re-layout of the weights; cached until the tensor is written again."""
        key = (weight.data_ptr(), weight._version, nb)
        if key != self._packed_key:
            plan = self._get_plan()
            nper = _lib.lic360_conv_plan_packed_floats(plan)
            if self._packed is None or self._packed.numel() != nper * nb:
                self._packed = torch.empty((nb, nper), dtype=torch.float32, device=weight.device)
            _chk(_lib.lic360_conv_pack(self._s(), plan, _p(weight), nb, _p(self._packed)))
            self._packed_key = key
        return self._packed

    def _args(self, x, weight, bias, act, batch):
        for t, nm in ((x, "input"), (weight, "weight"), (bias, "bias")):
            _check_in(t, self.device_, nm)
        if act is not None:
            _check_in(act, self.device_, "act")
        n, c, h, w = x.shape
        if c != self.channel_:
            raise Lic360Error("input has %d channels, op was built for %d" % (c, self.channel_))
        nb = int(weight.shape[0]) if batch else 1
        wshape = (self.nout_, self.channel_, self.kernel_size_, self.kernel_size_)
        if tuple(weight.shape) != ((nb,) + wshape if batch else wshape) or n % nb:
            raise Lic360Error("weight shape %s does not match the op" % (tuple(weight.shape),))
        return n, h, w, nb


class CconvEcOp(_ConvBase):
    def _run(self, x, weight, bias, act, batch):
        n, h, w, nb = self._args(x, weight, bias, act, batch)
        packed = self._pack(weight, nb)
        self._tops(x, [(n, self.nout_, h, w)])
        _chk(_lib.lic360_cconv_ec(self._s(), self._get_plan(), _p(x), _p(packed), _p(bias), _p(act), _p(self._top[0]), n, h, w, nb))
        return self._top

    def forward(self, x, weight, bias):
        return self._run(x, weight, bias, None, False)

    def forward_act(self, x, weight, bias, act):
        return self._run(x, weight, bias, act, False)

    def forward_batch(self, x, weight, bias):
        return self._run(x, weight, bias, None, True)

    def forward_act_batch(self, x, weight, bias, act):
        return self._run(x, weight, bias, act, True)


class CconvDcOp(_ConvBase):
    def _run(self, x, weight, bias, act, batch):
        n, h, w, nb = self._args(x, weight, bias, act, batch)
        self._plane_shape(x.shape)
        packed = self._pack(weight, nb)
        self._tops(x, [(n, self.nout_, h, w)])
        psum = self.plan_sum_
        self.plan_sum_ += 1
        if psum < h + w + self.ngroup_ - 2:
            start, ln = self._window(psum, h, w)
            if ln > 0:
                if psum == 0:
                    self._top[0].zero_()
                _chk(_lib.lic360_cconv_dc_plane(self._s(), self._get_plan(), _p(x), _p(packed), _p(bias), _p(act), _p(self._top[0]),
                                                n, h, w, nb, _p(self.index_mat_), _p(self._pidx_dev()), _p(self.plan_idx_mat_), psum))
        return self._top

    def forward(self, x, weight, bias):
        return self._run(x, weight, bias, None, False)

    def forward_act(self, x, weight, bias, act):
        return self._run(x, weight, bias, act, False)

    def forward_batch(self, x, weight, bias):
        return self._run(x, weight, bias, None, True)

    def forward_act_batch(self, x, weight, bias, act):
        return self._run(x, weight, bias, act, True)


# --------------------------------------------------------------------------------------- coder
class Coder(object):
    """Counterpart of the reference Coder (extension/coder.h:10-63): CPU tensors in, file out."""

    def __init__(self, name, file_value):
        self.fname = str(name)
        self.file_value = float(file_value)
        self._h = None

    def __del__(self):
        self._close()

    def _close(self):
        if getattr(self, "_h", None):
            _lib.lic360_coder_close(C.c_void_p(self._h))
            self._h = None

    def reset_fname(self, name):
        self.fname = str(name)

    def get_fname(self):
        return self.fname

    def start_encoder(self):
        self._close()
        self._h = _lib.lic360_coder_enc_open()

    def end_encoder(self):
        n = _lib.lic360_coder_enc_finish(C.c_void_p(self._h))
        if n < 0:
            raise Lic360Error("end_encoder without start_encoder")
        data = C.string_at(_lib.lic360_coder_bytes(C.c_void_p(self._h)), n)
        with open(self.fname, "wb") as f:
            f.write(data)
        self._close()

    def start_decoder(self):
        self._close()
        with open(self.fname, "rb") as f:
            data = f.read()
        self._h = _lib.lic360_coder_dec_open(data, C.c_long(len(data)))
        if not self._h:
            raise Lic360Error(_lib.lic360_last_error().decode())

    @staticmethod
    def _cpu_i32(t, name):
        if t.is_cuda or t.dtype != torch.int32:
            raise Lic360Error("%s must be a CPU int32 tensor" % name)
        return t.contiguous()

    def encodes(self, table, ncode, label, num):
        table, label = self._cpu_i32(table, "table"), self._cpu_i32(label, "label")
        _chk(_lib.lic360_coder_encode_slice(C.c_void_p(self._h), _p(table), int(ncode), _p(label), C.c_void_p(0), int(num)))

    def encodes_mask(self, table, ncode, label, mask, num):
        table, label = self._cpu_i32(table, "table"), self._cpu_i32(label, "label")
        mask = mask.to(torch.float32).contiguous()
        _chk(_lib.lic360_coder_encode_slice(C.c_void_p(self._h), _p(table), int(ncode), _p(label), _p(mask), int(num)))

    def decodes(self, table, ncode, num):
        table = self._cpu_i32(table, "table")
        out = torch.empty((table.shape[0],), dtype=torch.float32)
        _chk(_lib.lic360_coder_decode_slice(C.c_void_p(self._h), _p(table), int(ncode), C.c_void_p(0), _f(self.file_value), _p(out), int(num)))
        return out

    def decodes_mask(self, table, ncode, mask, num):
        table = self._cpu_i32(table, "table")
        mask = mask.to(torch.float32).contiguous()
        out = torch.empty((table.shape[0],), dtype=torch.float32)
        _chk(_lib.lic360_coder_decode_slice(C.c_void_p(self._h), _p(table), int(ncode), _p(mask), _f(self.file_value), _p(out), int(num)))
        return out

    def encode(self, table, ncode, tsum, symbol):      # Coder::my_encoder (extension/coder.cpp:12-20)
        t = table.to("cpu").to(torch.int32).contiguous().view(1, -1)
        lab = torch.tensor([int(symbol)], dtype=torch.int32)
        _chk(_lib.lic360_coder_encode_slice(C.c_void_p(self._h), _p(t), int(ncode), _p(lab), C.c_void_p(0), 1))

    def decode(self, table, ncode, tsum):               # Coder::my_decoder (extension/coder.cpp:21-29)
        t = table.to("cpu").to(torch.int32).contiguous().view(1, -1)
        out = torch.empty((1,), dtype=torch.float32)
        _chk(_lib.lic360_coder_decode_slice(C.c_void_p(self._h), _p(t), int(ncode), C.c_void_p(0), _f(self.file_value), _p(out), 1))
        return int(out[0])


# --------------------------------------------------------------------------------------- out-of-scope classes
def _out_of_scope(name):
    class _Stub(object):
        def __init__(self, *a, **k):
            raise NotImplementedError("%s is outside the hot path (SURVEY.md §2.1: quality metrics / viewer / training)" % name)
    _Stub.__name__ = name
    return _Stub


ProjectsOp = _out_of_scope("ProjectsOp")
MaskConstrainOp = _out_of_scope("MaskConstrainOp")
CppOp = _out_of_scope("CppOp")
ViewportOp = _out_of_scope("ViewportOp")
