"""ctypes front-end of the CPU oracle (oracle/liblic360_oracle.so) and, when built, of the
reference arithmetic coder (oracle/_ref/libref_ac.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblic360_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_ac.so")


def build(ref=True):
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    if ref:
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


def _load():
    if not os.path.exists(_LIB):
        build(ref=os.path.isdir("/root/reference"))
    return C.CDLL(_LIB)


lib = _load()
_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- code context
lib.orc_code_contex.argtypes = [C.c_int, C.c_int, _i, _i]
lib.orc_plane_len.argtypes = [C.c_int] * 4 + [_i]
lib.orc_plane_len.restype = C.c_int


def code_contex(H, W):
    idx = np.zeros(2 * H * W, np.int32)
    pidx = np.zeros(H + W, np.int32)
    lib.orc_code_contex(H, W, idx, pidx)
    return idx, pidx


def plane_len(p, G, H, W, pidx):
    return lib.orc_plane_len(p, G, H, W, pidx)


# ---------------------------------------------------------------- masked conv
_PF = C.POINTER(C.c_float)
lib.orc_cconv_ec.argtypes = [_f, _f, _f, _PF, _f] + [C.c_int] * 9
lib.orc_cconv_dc_plane.argtypes = [_f, _f, _f, _PF, _f] + [C.c_int] * 9 + [_i, _i, C.c_int]


def cconv_ec(x, w, b, act, ngroup, constrain):
    """x [N,C,H,W]; w [nb,nout,C,k,k] or [nout,C,k,k]; b/act [nb,nout] or [nout]."""
    x, w, b = f32(x), f32(w), f32(b)
    nb = w.shape[0] if w.ndim == 5 else 1
    nout, ksz = w.shape[-4], w.shape[-1]
    N, Cc, H, W = x.shape
    act = None if act is None else f32(act)
    out = np.empty((N, nout, H, W), np.float32)
    lib.orc_cconv_ec(x, w, b, _fp(act), out, N, Cc, H, W, nout, ngroup, ksz, constrain, nb)
    return out


def cconv_dc_plane(x, w, b, act, out, ngroup, constrain, idx, pidx, psum):
    x, w, b = f32(x), f32(w), f32(b)
    nb = w.shape[0] if w.ndim == 5 else 1
    nout, ksz = w.shape[-4], w.shape[-1]
    N, Cc, H, W = x.shape
    act = None if act is None else f32(act)
    assert out.dtype == np.float32 and out.shape == (N, nout, H, W)
    lib.orc_cconv_dc_plane(x, w, b, _fp(act), out, N, Cc, H, W, nout, ngroup, ksz, constrain, nb, idx, pidx, psum)
    return out


# ---------------------------------------------------------------- tile ops
lib.orc_tile_extract.argtypes = [_f, _f] + [C.c_int] * 6 + [_i, _i, C.c_int]
lib.orc_tile_extract.restype = C.c_int
lib.orc_tile_extract_batch.argtypes = [_f, _f] + [C.c_int] * 5 + [_i, _i, C.c_int]
lib.orc_tile_extract_batch.restype = C.c_int
lib.orc_tile_input.argtypes = [_f, _f] + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, _i, _i, C.c_int]
lib.orc_tile_add.argtypes = [_f, _f] + [C.c_int] * 5 + [_i, _i, C.c_int]


def tile_extract(x, out, ngroup, label, idx, pidx, psum):
    N, Cc, H, W = x.shape
    return lib.orc_tile_extract(f32(x), out, N, Cc, H, W, ngroup, int(label), idx, pidx, psum)


def tile_extract_batch(x, out, ngroup, idx, pidx, psum):
    N, Cc, H, W = x.shape
    return lib.orc_tile_extract_batch(f32(x), out, N, Cc, H, W, ngroup, idx, pidx, psum)


def tile_input(sym, out, N, G, H, W, bias, scale, rep, idx, pidx, psum):
    lib.orc_tile_input(f32(sym), out, N, G, H, W, bias, scale, rep, idx, pidx, psum)


def tile_add(y, x, ngroup, idx, pidx, psum):
    N, Cc, H, W = y.shape
    lib.orc_tile_add(y, f32(x), N, Cc, H, W, ngroup, idx, pidx, psum)


# ---------------------------------------------------------------- tables
lib.orc_gmm_table.argtypes = [_f, _f, _f, _f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
lib.orc_gmm_table_batch.argtypes = [_f, C.c_long, _f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
lib.orc_entropy_table.argtypes = [_f, _f, C.c_int, C.c_int, C.c_float]
lib.orc_entropy_gmm.argtypes = [_f] * 9 + [C.c_int, C.c_int]


def gmm_table(weight, delta, mean, tn, nstep=8, bias=3.5, total=65536.0, beta=1e-6):
    """weight/delta/mean: float32 [tn, ng] (mutated in place like the reference)."""
    ng = weight.shape[-1]
    out = np.zeros((tn, nstep + 1), np.float32)
    lib.orc_gmm_table(weight, delta, mean, out, tn, ng, nstep, bias, total, beta)
    return out


def gmm_table_batch(data, stride, tn, ng=3, nstep=8, bias=3.5, total=65536.0, beta=1e-6):
    out = np.zeros((max(tn, 1), nstep + 1), np.float32)
    lib.orc_gmm_table_batch(data, stride, out, tn, ng, nstep, bias, total, beta)
    return out[:tn]


def entropy_table(data, count, nstep, total=65536.0):
    out = np.zeros((max(count, 1), nstep + 1), np.float32)
    lib.orc_entropy_table(f32(data), out, count, nstep, total)
    return out[:count]


lib.orc_expf_v.argtypes = [_f, _f, C.c_long]
lib.orc_erff_v.argtypes = [_f, _f, C.c_long]


def expf(x):
    x = f32(np.atleast_1d(x))
    y = np.empty_like(x)
    lib.orc_expf_v(x.reshape(-1), y.reshape(-1), x.size)
    return y


def erff(x):
    x = f32(np.atleast_1d(x))
    y = np.empty_like(x)
    lib.orc_erff_v(x.reshape(-1), y.reshape(-1), x.size)
    return y


def entropy_gmm(weight, delta, mean, label):
    M, ng = weight.shape
    loss = np.zeros(M, np.float32)
    wd, dd, md = (np.zeros((M, ng), np.float32) for _ in range(3))
    ld = np.zeros((M, 1), np.float32)
    lib.orc_entropy_gmm(f32(weight), f32(delta), f32(mean), f32(label).reshape(-1), loss, wd, dd, md, ld.reshape(-1), M, ng)
    return loss, wd, dd, md, ld


# ---------------------------------------------------------------- layout / sphere / pointwise
lib.orc_context_reshape.argtypes = [_f, _f] + [C.c_int] * 5
lib.orc_contex_shift.argtypes = [_f, _f] + [C.c_int] * 6
lib.orc_sphere_pad.argtypes = [_f, _f] + [C.c_int] * 4
lib.orc_sphere_pad_inplace.argtypes = [_f] + [C.c_int] * 4
lib.orc_sphere_trim.argtypes = [_f] + [C.c_int] * 4
lib.orc_sphere_cut_edge.argtypes = [_f, _f] + [C.c_int] * 4
lib.orc_sphere_pad_backward.argtypes = [_f, _f] + [C.c_int] * 4
lib.orc_sphere_pad_backward_inplace.argtypes = [_f] + [C.c_int] * 4
lib.orc_sphere_cut_edge_backward.argtypes = [_f, _f] + [C.c_int] * 4
lib.orc_sphere_lat_scale.argtypes = [_f, _f, _f] + [C.c_int] * 4
lib.orc_imp_map.argtypes = [_f, _f, _f, _PF] + [C.c_int] * 5
lib.orc_imp_map_constrain.argtypes = [_f, C.c_int, C.c_int, C.c_float, C.c_float]
lib.orc_imp2mask.argtypes = [_f, _f] + [C.c_int] * 5
lib.orc_mask_constrain.argtypes = [_f] + [C.c_int] * 5
lib.orc_scale.argtypes = [_f, _f, C.c_long, C.c_float, C.c_float]
lib.orc_quant.argtypes = [_f, _f, _f, _PF, _f] + [C.c_int] * 5
lib.orc_dquant.argtypes = [_f, _f, _f, _f] + [C.c_int] * 5
lib.orc_dtow.argtypes = [_f, _f] + [C.c_int] * 6


def context_reshape(x, ngroup):
    N, Cc, H, W = x.shape
    out = np.empty((N * H * W * ngroup, Cc // ngroup), np.float32)
    lib.orc_context_reshape(f32(x), out, N, Cc, H, W, ngroup)
    return out


def contex_shift(x, cpn, inv):
    N, Cc, H, W = x.shape
    G = Cc // cpn
    Ho = H - W - G + 2 if inv else H + W + G - 2
    out = np.empty((N, Cc, Ho, W), np.float32)
    lib.orc_contex_shift(f32(x), out, N, Cc, H, W, cpn, int(inv))
    return out


def sphere_pad(x, pad):
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, H + 2 * pad, W + 2 * pad), np.float32)
    lib.orc_sphere_pad(f32(x), out, N * Cc, H, W, pad)
    return out


def sphere_pad_inplace(x, pad):
    N, Cc, H, W = x.shape
    lib.orc_sphere_pad_inplace(x, N * Cc, H, W, pad)
    return x


def sphere_trim(x, pad):
    N, Cc, H, W = x.shape
    lib.orc_sphere_trim(x, N * Cc, H, W, pad)
    return x


def sphere_cut_edge(x, pad):
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, H - 2 * pad, W - 2 * pad), np.float32)
    lib.orc_sphere_cut_edge(f32(x), out, N * Cc, H, W, pad)
    return out


def sphere_pad_backward(top_diff, pad):
    N, Cc, Ho, Wo = top_diff.shape
    out = np.empty((N, Cc, Ho - 2 * pad, Wo - 2 * pad), np.float32)
    lib.orc_sphere_pad_backward(out, f32(top_diff), N * Cc, Ho - 2 * pad, Wo - 2 * pad, pad)
    return out


def sphere_pad_backward_inplace(diff, pad):
    N, Cc, Hp, Wp = diff.shape
    lib.orc_sphere_pad_backward_inplace(diff, N * Cc, Hp, Wp, pad)
    return diff


def sphere_cut_edge_backward(top_diff, pad):
    N, Cc, Ho, Wo = top_diff.shape
    out = np.empty((N, Cc, Ho + 2 * pad, Wo + 2 * pad), np.float32)
    lib.orc_sphere_cut_edge_backward(out, f32(top_diff), N * Cc, Ho + 2 * pad, Wo + 2 * pad, pad)
    return out


def sphere_lat_scale(x, weight, npart):
    N, Cc, H, W = x.shape
    out = np.empty_like(x, dtype=np.float32)
    lib.orc_sphere_lat_scale(f32(x), f32(weight).reshape(-1), out, N * Cc, H, W, npart)
    return out


def imp_map(x, imp, levels, want_mask=True):
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, H, W), np.float32)
    mask = np.empty((N, Cc, H, W), np.float32) if want_mask else None
    lib.orc_imp_map(f32(x), f32(imp), out, _fp(mask), N, Cc, H, W, levels)
    return out, mask


def imp_map_constrain(N, H, rt, sc):
    out = np.empty((N, 1, H), np.float32)
    lib.orc_imp_map_constrain(out.reshape(-1), N, H, rt, sc)
    return out


lib.orc_imp_map_alpha.argtypes = [_f, C.c_int, C.c_float, C.c_float]
lib.orc_imp_map_backward.argtypes = [_f, _f, _f, _f, _f, _f] + [C.c_int] * 6 + [C.c_float]


def imp_map_alpha(H, alpha, sw):
    out = np.empty(H, np.float32)
    lib.orc_imp_map_alpha(out, H, alpha, sw)
    return out


def imp_map_backward(top_diff, imp, sphere_constrain, alpha_t, levels, imp_kernel, gamma):
    N, Cc, H, W = top_diff.shape
    dd, di = np.empty((N, Cc, H, W), np.float32), np.empty((N, 1, H, W), np.float32)
    lib.orc_imp_map_backward(f32(top_diff), f32(imp), f32(sphere_constrain).reshape(-1), f32(alpha_t), dd, di, N, Cc, H, W, levels, imp_kernel, gamma)
    return dd, di


def imp2mask(x, levels, channels):
    N, _, H, W = x.shape
    out = np.empty((N, channels, H, W), np.float32)
    lib.orc_imp2mask(f32(x), out, N, channels, H, W, channels // levels)
    return out


def mask_constrain(w, ngroup, constrain):
    """-> masked copy of w [nout, channel, k, k]"""
    out = np.ascontiguousarray(w, dtype=np.float32).copy()
    lib.orc_mask_constrain(out, out.shape[0], out.shape[1], out.shape[2], ngroup, constrain)
    return out


def scale(x, bias, scale_):
    out = np.empty_like(x, dtype=np.float32)
    lib.orc_scale(f32(x), out, x.size, bias, scale_)
    return out


def quant(x, weight_b):
    N, Cc, H, W = x.shape
    levels = weight_b.shape[1]
    top = np.empty((N, Cc, H, W), np.float32)
    qidx = np.empty((N, Cc, H, W), np.float32)
    count = np.zeros((Cc, levels), np.float32)
    lib.orc_quant(f32(x), f32(weight_b), top, _fp(qidx), count, N, Cc, H, W, levels)
    return top, qidx, count


lib.orc_quant_update_weight.argtypes = [_f, _f, C.c_int, C.c_int, C.c_float]
lib.orc_quant_backward.argtypes = [_f, C.c_void_p, _f, _f, _f, _f, _f, _f] + [C.c_int] * 5 + [C.c_float]


def quant_update_weight(weight_b, ncount, weight_decay):
    w, c = np.ascontiguousarray(weight_b, np.float32).copy(), np.ascontiguousarray(ncount, np.float32).copy()
    lib.orc_quant_update_weight(w, c, w.shape[0], w.shape[1], weight_decay)
    return w, c


def quant_backward(top_diff0, top_diff1, bottom_data, top_data, qidx, weight_b, top_alpha):
    N, Cc, H, W = bottom_data.shape
    dd, wd = np.empty((N, Cc, H, W), np.float32), np.empty(weight_b.shape, np.float32)
    lib.orc_quant_backward(f32(top_diff0), _fp(None if top_diff1 is None else f32(top_diff1)), f32(bottom_data), f32(top_data), f32(qidx), f32(weight_b),
                           dd, wd, N, Cc, H, W, weight_b.shape[1], top_alpha)
    return dd, wd


def dquant(x, mask, weight_b):
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, H, W), np.float32)
    lib.orc_dquant(f32(x), f32(mask), f32(weight_b), out, N, Cc, H, W, weight_b.shape[1])
    return out


def dtow(x, stride, d2w):
    N, Cc, H, W = x.shape
    s2 = stride * stride
    shp = (N, Cc // s2, H * stride, W * stride) if d2w else (N, Cc * s2, H // stride, W // stride)
    out = np.empty(shp, np.float32)
    lib.orc_dtow(f32(x), out, N, Cc, H, W, stride, int(d2w))
    return out


# ---------------------------------------------------------------- arithmetic coder
lib.orc_ac_enc_open.restype = C.c_void_p
lib.orc_ac_dec_open.restype = C.c_void_p
lib.orc_ac_dec_open.argtypes = [_u8, C.c_size_t]
lib.orc_ac_close.argtypes = [C.c_void_p]
lib.orc_ac_error.argtypes = [C.c_void_p]
lib.orc_ac_error.restype = C.c_int
lib.orc_ac_encode_slice.argtypes = [C.c_void_p, _i, C.c_int, _i, _PF, C.c_int]
lib.orc_ac_enc_finish.argtypes = [C.c_void_p]
lib.orc_ac_enc_finish.restype = C.c_size_t
lib.orc_ac_bytes.argtypes = [C.c_void_p]
lib.orc_ac_bytes.restype = C.POINTER(C.c_uint8)
lib.orc_ac_decode_slice.argtypes = [C.c_void_p, _i, C.c_int, _PF, C.c_float, _f, C.c_int]


class Encoder:
    def __init__(self):
        self.h = lib.orc_ac_enc_open()

    def encode(self, table, ncode, label, mask, num):
        table = np.ascontiguousarray(table, np.int32)
        label = np.ascontiguousarray(label, np.int32)
        mask = None if mask is None else f32(mask)
        lib.orc_ac_encode_slice(self.h, table.reshape(-1), ncode, label.reshape(-1), _fp(mask), num)
        if lib.orc_ac_error(self.h):
            raise RuntimeError("oracle AC encoder error %d" % lib.orc_ac_error(self.h))

    def finish(self):
        n = lib.orc_ac_enc_finish(self.h)
        data = bytes(bytearray(lib.orc_ac_bytes(self.h)[:n]))
        lib.orc_ac_close(self.h)
        self.h = None
        return data


class Decoder:
    def __init__(self, data):
        arr = np.frombuffer(data, np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        self.h = lib.orc_ac_dec_open(arr, len(data))

    def decode(self, table, ncode, mask, num, file_value=3.5, size=None):
        table = np.ascontiguousarray(table, np.int32)
        mask = None if mask is None else f32(mask)
        out = np.zeros(size if size is not None else max(num, 1), np.float32)
        lib.orc_ac_decode_slice(self.h, table.reshape(-1), ncode, _fp(mask), file_value, out, num)
        if lib.orc_ac_error(self.h):
            raise RuntimeError("oracle AC decoder error %d" % lib.orc_ac_error(self.h))
        return out

    def close(self):
        if self.h:
            lib.orc_ac_close(self.h)
            self.h = None


# ---------------------------------------------------------------- reference coder (oracle/_ref)
def have_ref():
    return os.path.exists(_REF)


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF)
        _ref.ref_ac_encode.argtypes = [_i, C.c_int, _i, _PF, C.c_long, _u8, C.c_long]
        _ref.ref_ac_encode.restype = C.c_long
        _ref.ref_ac_decode.argtypes = [_u8, C.c_long, _i, C.c_int, _PF, C.c_float, C.c_long, _f]
        _ref.ref_ac_decode.restype = C.c_int
    return _ref


def ref_encode(tables, ncode, labels, mask):
    tables = np.ascontiguousarray(tables, np.int32)
    labels = np.ascontiguousarray(labels, np.int32)
    mask = None if mask is None else f32(mask)
    num = labels.size
    cap = num * 8 + 64
    out = np.zeros(cap, np.uint8)
    n = ref_lib().ref_ac_encode(tables.reshape(-1), ncode, labels.reshape(-1), _fp(mask), num, out, cap)
    if n < 0:
        raise RuntimeError("reference coder failed (%d)" % n)
    return out[:n].tobytes()


def ref_decode(data, tables, ncode, mask, num, file_value=3.5):
    tables = np.ascontiguousarray(tables, np.int32)
    mask = None if mask is None else f32(mask)
    arr = np.frombuffer(data, np.uint8).copy()
    out = np.zeros(num, np.float32)
    rc = ref_lib().ref_ac_decode(arr, arr.size, tables.reshape(-1), ncode, _fp(mask), file_value, num, out)
    if rc:
        raise RuntimeError("reference decoder failed (%d)" % rc)
    return out


# ---------------------------------------------------------------- f3 viewport projection
lib.orc_projects_tf.argtypes = [_f, C.c_int, C.c_int, _f, _f, C.c_float, C.c_int, C.c_int]
lib.orc_projects_forward.argtypes = [_f, _f, _f] + [C.c_int] * 5
lib.orc_projects_backward.argtypes = [_f, _f, _f, _f] + [C.c_int] * 5


def projects_tf(h_out, w_out, theta, phi, fov, height, width):
    tf = np.empty((14, h_out * w_out, 2), np.float32)
    lib.orc_projects_tf(tf, h_out, w_out, f32(theta), f32(phi), fov, height, width)
    return tf


def projects_forward(x, tf, h_out, w_out, nearest=False):
    N, Cc, H, W = x.shape
    out = np.empty((14 * N, Cc, h_out, w_out), np.float32)
    lib.orc_projects_forward(f32(x), f32(tf), out, N * Cc, H, W, h_out * w_out, int(nearest))
    return out


def projects_backward(g, tf, N, Cc, H, W, nearest=False):
    inner = g.shape[-1] * g.shape[-2]
    d, c = np.empty((N, Cc, H, W), np.float32), np.empty((N, Cc, H, W), np.float32)
    lib.orc_projects_backward(f32(g), f32(tf), d, c, N * Cc, H, W, inner, int(nearest))
    return d, c


# ---------------------------------------------------------------- CppOp (Craster parabolic projection)
lib.orc_cpp_forward.argtypes = [_f, _f, C.c_void_p, C.c_int, C.c_int, C.c_int]


def cpp_forward(x, want_mask=False):
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, H, W), np.float32)
    mask = np.empty((N, Cc, H, W), np.float32) if want_mask else None
    lib.orc_cpp_forward(f32(x), out, _fp(mask), N * Cc, H, W)
    return out, mask


# ---------------------------------------------------------------- ViewportOp
lib.orc_viewport_forward.argtypes = [_f] * 7 + [C.c_int] * 6 + [C.c_float]
lib.orc_viewport_xy.argtypes = [_f, _f, _f, C.c_int, C.c_int, C.c_int, C.c_float]


def viewport_forward(x, theta_phi, ho, wo, fov_deg):
    N, Cc, H, W = x.shape
    out, r0 = np.empty((N, Cc, ho, wo), np.float32), np.empty((N, ho, wo, 3), np.float32)
    rota, rays, tf = np.empty((N, 9), np.float32), np.empty((N, ho, wo, 3), np.float32), np.empty((N, ho, wo, 2), np.float32)
    lib.orc_viewport_forward(f32(x), f32(theta_phi), out, r0, rota, rays, tf, N, Cc, H, W, ho, wo, fov_deg)
    return out, r0, rota, rays, tf


def viewport_xy(theta_phi_next, rota, ho, wo, fov_deg):
    n = theta_phi_next.shape[0]
    xy = np.empty((n, 2), np.float32)
    lib.orc_viewport_xy(f32(theta_phi_next), f32(rota), xy, n, ho, wo, fov_deg)
    return xy


# ---------------------------------------------------------------- f1: transform blocks (test/model_zoo.py:8-105,145-170; GDN.py:66-100)
lib.orc_conv2d.argtypes = [_f, _f, _PF, _f] + [C.c_int] * 8
lib.orc_prelu.argtypes = [_f, _f, _f, C.c_int, C.c_int, C.c_long]
lib.orc_gdn.argtypes = [_f, _f, _f, _f, C.c_int, C.c_int, C.c_long, C.c_int]


def conv2d(x, w, b, stride=1, pad=0):
    x, w = f32(x), f32(w)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    b = None if b is None else f32(b)
    out = np.empty((N, Cout, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1), np.float32)
    lib.orc_conv2d(x, w, _fp(b), out, N, Cin, H, W, Cout, k, stride, pad)
    return out


def prelu(x, a):
    x = f32(x)
    out = np.empty_like(x)
    lib.orc_prelu(x, f32(a), out, x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
    return out


def gdn(x, gamma_eff, beta_eff, inverse=False):
    x = f32(x)
    out = np.empty_like(x)
    lib.orc_gdn(x, f32(gamma_eff), f32(beta_eff), out, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], int(inverse))
    return out


def gdn_effective(gamma, beta, pedestal, beta_bound, gamma_bound):
    """the reparametrisation of GDN.py:81-86: lower bound, square, minus the pedestal (fp32, in that order)"""
    g = np.maximum(f32(gamma), np.float32(gamma_bound))
    bt = np.maximum(f32(beta), np.float32(beta_bound))
    return (g * g - np.float32(pedestal)).astype(np.float32), (bt * bt - np.float32(pedestal)).astype(np.float32)


class blocks:
    """The reference's transform blocks over the oracle's ops.  p: dict of numpy parameters with the block's state_dict keys
    (conv1.weight, relu1.weight, relu2.gamma / relu2.beta + its `gdn` constants ...).  Inputs carry a 2-cell sphere apron."""

    @staticmethod
    def _gdn(y, p, pre, inverse):
        ge, be = gdn_effective(p[pre + ".gamma"], p[pre + ".beta"], p[pre + ".pedestal"], p[pre + ".beta_bound"], p[pre + ".gamma_bound"])
        return gdn(y, ge, be, inverse)

    @staticmethod
    def residual(x, p):                                   # model_zoo.py:8-23
        y = sphere_pad_inplace(f32(x).copy(), 2)
        y = prelu(conv2d(y, p["conv1.weight"], p["conv1.bias"]), p["relu1.weight"])
        y = prelu(conv2d(y, p["conv2.weight"], p["conv2.bias"], 1, 1), p["relu2.weight"])
        return sphere_trim((sphere_pad_inplace(f32(x).copy(), 2) + conv2d(y, p["conv3.weight"], p["conv3.bias"])).astype(np.float32), 2)

    @staticmethod
    def residual_v2(x, p):                                # model_zoo.py:48-64
        y = sphere_pad_inplace(f32(x).copy(), 2)
        y = sphere_trim(prelu(conv2d(y, p["conv1.weight"], p["conv1.bias"], 1, 1), p["relu1.weight"]), 1)
        y = sphere_trim(prelu(conv2d(y, p["conv2.weight"], p["conv2.bias"], 1, 1), p["relu2.weight"]), 2)
        return (sphere_pad_inplace(f32(x).copy(), 2) + y).astype(np.float32)

    @staticmethod
    def residual_down(x, p):                              # model_zoo.py:66-95, hidden = True
        t = conv2d(x, p["short_cut.weight"], p["short_cut.bias"], 2, 2)
        y = sphere_pad_inplace(f32(x).copy(), 2)
        y = sphere_trim(prelu(conv2d(y, p["conv1.weight"], p["conv1.bias"], 2, 3), p["relu1.weight"]), 2)
        y = sphere_pad_inplace(y, 2)
        y = blocks._gdn(conv2d(y, p["conv2.weight"], p["conv2.bias"], 1, 1), p, "relu2", False)
        return sphere_trim((t + y).astype(np.float32), 2)

    @staticmethod
    def residual_up(x, p):                                # model_zoo.py:145-170
        y = sphere_pad_inplace(f32(x).copy(), 2)
        b = prelu(conv2d(y, p["conv1.weight"], p["conv1.bias"]), p["relu1.weight"])
        b = sphere_pad_inplace(sphere_trim(dtow(b, 2, True), 2), 2)
        b = blocks._gdn(conv2d(b, p["conv2.weight"], p["conv2.bias"], 1, 1), p, "relu2", True)
        s = dtow(conv2d(sphere_cut_edge(y, 1), p["short_cut.weight"], p["short_cut.bias"]), 2, True)
        return sphere_trim((b + s).astype(np.float32), 2)
