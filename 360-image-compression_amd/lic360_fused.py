"""FusedCodec -- host wrapper of the device-resident latent codec (C ABI lic360_codec_*, csrc/codec_fused.hip).

Replaces the EntEncoderFast / EntDecoder pair (test/lic360_demo.py:95-141,191-238) for batches of images:
inputs, activations, CDF tables, coder state and bitstreams all stay in HBM; the host only queues
kernels.  Bitstreams are byte-identical to the per-plane drivers in lic360_codec.py and to the oracle.
torch is used for device buffers and streams only.
"""
import ctypes as C

import torch

import lic360
from lic360 import _lib, _chk, _p, _stream, Lic360Error



class FusedCodec(object):
    def __init__(self, ngroup, h, w, max_batch, device=0, cap_bytes=None):
        self.G, self.H, self.W, self.maxB, self.device = int(ngroup), int(h), int(w), int(max_batch), int(device)
        # worst case of this coder is < 2.1 bytes/symbol (16-bit CDF, frequency >= 1); 1 byte/symbol is ample for
        # any real table and overflow is reported through err[] rather than written out of bounds
        self.cap = int(cap_bytes) if cap_bytes else max(4096, self.G * self.H * self.W)
        self.cap = (self.cap + 3) // 4 * 4              # streams start word-aligned (the device bit reader fetches words)
        self._h = C.c_void_p(0)
        with torch.cuda.device(self.device):
            _chk(_lib.lic360_codec_create(self.G, self.H, self.W, self.maxB, C.byref(self._h)))
        dev = "cuda:%d" % self.device
        self.bytes = torch.zeros((self.maxB, self.cap), dtype=torch.uint8, device=dev)
        self.nbytes = torch.zeros((self.maxB,), dtype=torch.int32, device=dev)
        self.err = torch.zeros((self.maxB,), dtype=torch.int32, device=dev)
        self.code_out = torch.zeros((self.maxB, self.G, self.H, self.W), dtype=torch.float32, device=dev)

    def __del__(self):
        try:
            if self._h:
                _lib.lic360_codec_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------------
    def set_layer(self, layer, weight, bias, act=None):
        for t in (weight, bias) + ((act,) if act is not None else ()):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise Lic360Error("codec parameters must be contiguous float32 device tensors")
        _chk(_lib.lic360_codec_set_layer(_stream(self.device), self._h, int(layer), _p(weight), _p(bias), _p(act)))

    def load_from_driver(self, drv):
        """Take the 12 batched layers of an EntEncoderFast / EntDecoder (state_dict keys net.N.*)."""
        mods = [drv.net[0]]
        for i in range(1, 6):
            mods += [drv.net[i].conv1, drv.net[i].conv2]
        mods.append(drv.net[6])
        for i, m in enumerate(mods):
            self.set_layer(i, m.weight.data.contiguous(), m.bias.data.contiguous(), None if m.relu is None else m.relu.data.contiguous())

    def load_layers(self, layers):
        """layers: list of 12 dicts with numpy/torch 'w' [3,nout,C,5,5], 'b' [3,nout], 'a' [3,nout] or None."""
        dev = "cuda:%d" % self.device
        to = lambda a: None if a is None else torch.as_tensor(a).to(dev).contiguous()
        for i, l in enumerate(layers):
            self.set_layer(i, to(l["w"]), to(l["b"]), to(l["a"]))
        torch.cuda.synchronize(self.device)

    # ---- codec ---------------------------------------------------------------------------------------
    def _check(self, t, name):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape[1:]) == (self.G, self.H, self.W)
                and 0 < t.shape[0] <= self.maxB):
            raise Lic360Error("%s must be a contiguous float32 device tensor [b<=%d,%d,%d,%d]" % (name, self.maxB, self.G, self.H, self.W))

    def encode_async(self, code, mask):
        """Queue the encode of a batch; results land in self.bytes / self.nbytes / self.err (device)."""
        self._check(code, "code")
        self._check(mask, "mask")
        b = code.shape[0]
        _chk(_lib.lic360_codec_encode(_stream(self.device), self._h, _p(code), _p(mask), b, _p(self.bytes), C.c_long(self.cap),
                                      _p(self.nbytes), _p(self.err)))
        return b

    def decode_async(self, mask, b=None, bytes_dev=None, nbytes_dev=None, gate=None):
        """gate: what FusedImpCodec.decode_masked_async returned IN THIS STEP -- `mask` (the very buffer handed to it) is then being
        filled plane by plane on another stream and every plane's table kernel waits for the event that covers it.  A gate is a
        one-time ticket: the map's masked decode must be enqueued first, and a gate of an earlier step (or one used twice) raises
        instead of decoding against a stale mask.  The caller orders the NEXT masked decode into the same buffer after this decode."""
        self._check(mask, "mask")
        b = mask.shape[0] if b is None else b
        bd = self.bytes if bytes_dev is None else bytes_dev
        nd = self.nbytes if nbytes_dev is None else nbytes_dev
        if gate is None:
            _chk(_lib.lic360_codec_decode(_stream(self.device), self._h, _p(bd), C.c_long(bd.shape[1]), _p(nd), _p(mask), b,
                                          _p(self.code_out), _p(self.err)))
        else:
            map_codec, generation = gate                                  # (the tuple keeps the map codec, hence its events, alive)
            _chk(_lib.lic360_codec_decode_gated(_stream(self.device), self._h, _p(bd), C.c_long(bd.shape[1]), _p(nd), _p(mask), b,
                                                _p(self.code_out), _p(self.err), map_codec._h, C.c_long(generation)))
        return self.code_out[:b]

    def encode(self, code, mask):
        """-> list of `bytes`, one bitstream per image (what the reference writes to `<code>`)."""
        b = self.encode_async(code, mask)
        nb = self.nbytes[:b].cpu().tolist()
        er = self.err[:b].cpu().tolist()
        if any(er):
            raise Lic360Error("arithmetic encoder fault / capacity overflow: %s" % er)
        host = self.bytes[:b].cpu()
        return [bytes(host[i, :nb[i]].numpy().tobytes()) for i in range(b)]

    def decode(self, streams, mask):
        b = len(streams)
        host = torch.zeros((self.maxB, self.cap), dtype=torch.uint8)
        nb = torch.zeros((self.maxB,), dtype=torch.int32)
        for i, s in enumerate(streams):
            if len(s) > self.cap:
                raise Lic360Error("bitstream %d longer than the codec capacity" % i)
            host[i, :len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8)
            nb[i] = len(s)
        self.bytes.copy_(host)
        self.nbytes.copy_(nb)
        out = self.decode_async(mask, b).clone()
        if int(self.err[:b].abs().sum().item()):
            raise Lic360Error("arithmetic decoder fault (corrupt stream?): %s" % self.err[:b].cpu().tolist())
        return out

    def set_coder(self, mode):
        """where the serial coder phases run: "device" (one wave per image), "host" (one host thread per image, <= 64 images per call), "auto"
        (default: host for calls of at most 8 images -- the latency regime)"""
        _chk(_lib.lic360_codec_set_coder(self._h, {"device": 0, "host": 1, "auto": 2}[mode]))

    # ---- dead-cone skip (csrc/need.h): statistics and test hooks ---------------------------------------
    def skip_active(self):
        """0: this codec computes every output (generic kernels or LIC360_NOSKIP); 1: the encode-order launches skip dead (tile, group block)
        pairs; 2: the decode-order launches of batches of >= 16 images (8 | batch) skip dead rows as well"""
        act = C.c_int(0)
        _chk(_lib.lic360_codec_skip_stats(self._h, -1, None, C.byref(act)))
        return act.value

    def skip_stats(self, enable=True, read=False):
        """enable: count what the following encodes / decodes execute.  read: -> (enc [12, 64] live (tile, group block) pairs per (layer,
        group block), dec [12, 64] stored cells per (layer, group)) since the last read, as numpy uint64 arrays; reading clears."""
        import numpy as np
        out = np.zeros((2, 12, 64), np.uint64) if read else None
        _chk(_lib.lic360_codec_skip_stats(self._h, int(bool(enable)), out.ctypes.data_as(C.c_void_p) if read else None, None))
        return (out[0], out[1]) if read else None

    def debug_fill(self, value):
        _chk(_lib.lic360_codec_debug_fill(_stream(self.device), self._h, C.c_float(float(value))))

    def debug_lists(self, which):
        """what the last encode / decode scheduled (which: see lic360_codec_debug_lists) as a flat numpy array + the list capacity"""
        import numpy as np
        P = self.H + self.W + self.G - 2
        cap = C.c_int(0)
        probe = np.zeros(1, np.int8)
        _chk(_lib.lic360_codec_debug_lists(self._h, int(which), probe.ctypes.data_as(C.c_void_p), C.c_long(0), C.byref(cap)))
        n = {0: self.maxB * 12 * self.H * self.W, 1: 12 * 8 * 4, 2: 12 * 8 * cap.value * 4, 3: 12 * P * 8 * 4, 4: 12 * P * 8 * cap.value * 16}[int(which)]
        buf = np.zeros(n, np.int8)
        _chk(_lib.lic360_codec_debug_lists(self._h, int(which), buf.ctypes.data_as(C.c_void_p), C.c_long(n), None))
        return (buf if which == 0 else buf.view(np.uint32 if which == 4 else np.int32)), cap.value

    # ---- timing hooks (bench.py) ---------------------------------------------------------------------
    def profile(self, on=True):
        _chk(_lib.lic360_codec_profile_enable(self._h, int(on)))

    def profile_read(self):
        """-> {kernel class: (total ms, launches)} since the last call (classes: lic360_codec_profile_classes)."""
        names = _lib.lic360_codec_profile_classes().decode().split(",")
        ms = (C.c_double * len(names))()
        cnt = (C.c_long * len(names))()
        _chk(_lib.lic360_codec_profile_read(self._h, len(names), ms, cnt))
        return {n: (ms[i], cnt[i]) for i, n in enumerate(names)}


class FusedImpCodec(object):
    """Device-resident importance-map stream: ImpEntEncoderFast + ImpEntDecoder (test/lic360_demo.py:143-189, 241-290) for a
    batch of maps.  levels: float32 [b,1,h,w] with values in {0..nsym-1}; bitstreams == the reference's `<code>_imp` files."""

    def __init__(self, h, w, max_batch, hidden_channels=144, nsym=49, device=0, cap_bytes=None):
        self.H, self.W, self.maxB, self.device = int(h), int(w), int(max_batch), int(device)
        self.cpg, self.nsym = int(hidden_channels), int(nsym)
        self.cap = int(cap_bytes) if cap_bytes else max(4096, 2 * self.H * self.W)
        self.cap = (self.cap + 3) // 4 * 4
        self._h = C.c_void_p(0)
        with torch.cuda.device(self.device):
            _chk(_lib.lic360_impcodec_create(self.H, self.W, self.cpg, self.nsym, self.maxB, C.byref(self._h)))
        dev = "cuda:%d" % self.device
        self.bytes = torch.zeros((self.maxB, self.cap), dtype=torch.uint8, device=dev)
        self.nbytes = torch.zeros((self.maxB,), dtype=torch.int32, device=dev)
        self.err = torch.zeros((self.maxB,), dtype=torch.int32, device=dev)
        self.levels_out = torch.zeros((self.maxB, 1, self.H, self.W), dtype=torch.float32, device=dev)

    def __del__(self):
        try:
            if self._h:
                _lib.lic360_impcodec_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def load_layers(self, layers):
        """layers: 12 dicts with numpy/torch 'w' [nout,C,5,5], 'b' [nout], 'a' [nout] or None."""
        dev = "cuda:%d" % self.device
        to = lambda a: None if a is None else torch.as_tensor(a).to(dev).contiguous()
        keep = []
        for i, l in enumerate(layers):
            w, b, a = to(l["w"]), to(l["b"]), to(l["a"])
            keep.append((w, b, a))
            _chk(_lib.lic360_impcodec_set_layer(_stream(self.device), self._h, i, _p(w), _p(b), _p(a)))
        torch.cuda.synchronize(self.device)

    def load_from_driver(self, drv):
        """drv: an ImpEntEncoderFast / ImpEntDecoder whose net.* parameters were filled by cast_imp_entropy_parameter."""
        mods = [drv.net[0]]
        for i in range(1, 6):
            mods += [drv.net[i].conv1, drv.net[i].conv2]
        mods.append(drv.net[6])
        self.load_layers([dict(w=m.weight.data, b=m.bias.data, a=None if m.relu is None else m.relu.data) for m in mods])

    def _check(self, t):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape[1:]) == (1, self.H, self.W) and 0 < t.shape[0] <= self.maxB):
            raise Lic360Error("levels must be a contiguous float32 device tensor [b<=%d,1,%d,%d]" % (self.maxB, self.H, self.W))

    def encode_async(self, levels):
        self._check(levels)
        b = levels.shape[0]
        _chk(_lib.lic360_impcodec_encode(_stream(self.device), self._h, _p(levels), b, _p(self.bytes), C.c_long(self.cap), _p(self.nbytes), _p(self.err)))
        return b

    def decode_async(self, b):
        _chk(_lib.lic360_impcodec_decode(_stream(self.device), self._h, _p(self.bytes), C.c_long(self.cap), _p(self.nbytes), int(b),
                                         _p(self.levels_out), _p(self.err)))
        return self.levels_out[:b]

    def decode_masked_async(self, b, mask_out, mask_channels=192, stride=2):
        """decode_async + the latent codec's mask Dtow(stride)(Imp2mask(levels)) into `mask_out`
        [b, mask_channels / stride^2, stride h, stride w], refreshed after every plane.  Returns the gate (this codec, ticket)
        to hand to FusedCodec.decode_async(mask_out, ..., gate=...) ON ANOTHER STREAM, AFTER this call: the latent decode then
        runs behind this one instead of after it (include/lic360_hip.h: lic360_impcodec_decode_masked, ordering contract)."""
        want = (int(b), mask_channels // (stride * stride), stride * self.H, stride * self.W)
        if not (mask_out.is_cuda and mask_out.dtype == torch.float32 and mask_out.is_contiguous() and tuple(mask_out.shape) == want):
            raise Lic360Error("mask_out must be a contiguous float32 device tensor %s" % (want,))
        gen = C.c_long(0)
        _chk(_lib.lic360_impcodec_decode_masked(_stream(self.device), self._h, _p(self.bytes), C.c_long(self.cap), _p(self.nbytes), int(b),
                                                _p(self.levels_out), _p(self.err), _p(mask_out), int(mask_channels), int(stride),
                                                C.byref(gen)))
        return (self, gen.value)

    def encode(self, levels):
        b = self.encode_async(levels)
        nb = self.nbytes[:b].cpu().tolist()
        er = self.err[:b].cpu().tolist()
        if any(er):
            raise Lic360Error("arithmetic encoder fault / capacity overflow: %s" % er)
        host = self.bytes[:b].cpu()
        return [bytes(host[i, :nb[i]].numpy().tobytes()) for i in range(b)]

    def decode(self, streams):
        b = len(streams)
        host = torch.zeros((self.maxB, self.cap), dtype=torch.uint8)
        nb = torch.zeros((self.maxB,), dtype=torch.int32)
        for i, s in enumerate(streams):
            if len(s) > self.cap:
                raise Lic360Error("bitstream %d longer than the codec capacity" % i)
            host[i, :len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8)
            nb[i] = len(s)
        self.bytes.copy_(host)
        self.nbytes.copy_(nb)
        out = self.decode_async(b).clone()
        if int(self.err[:b].abs().sum().item()):
            raise Lic360Error("arithmetic decoder fault (corrupt stream?): %s" % self.err[:b].cpu().tolist())
        return out
