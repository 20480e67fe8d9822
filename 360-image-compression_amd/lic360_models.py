"""Analysis / synthesis transforms of the LIC360 codec on this backend: the networks that turn a 512x1024 ERP image
into the (quantised latent, importance mask, importance map) triple the entropy coder consumes, and back.

Architecture and parameter names follow the reference's inference models (test/model_zoo.py:8-205, 334-379) so that its checkpoints
load with `load_state_dict`: the same module tree (`encoder.net.N...`, `decoder.net.N...`, `quant`, `imp`), built here from a few
generic pieces.  When no gradient is recorded, the 3x3 stride-1 convolutions of 96 / 192 channels on maps large enough to fill the chip run
on this package's own kernel (csrc/conv3x3_kernels.hip, `lic360.sconv3x3`): fp32 MFMA, the sphere apron read by index in its tile loader
(no in-place SpherePad in front of it), bias + PReLU + the residual add in its epilogue, the SphereTrim behind it as its output window.
The stride-2 and 1x1 convolutions, small maps and every recording (training) pass are library work (torch -> MIOpen); native around them:
sphere pad / trim / cut-edge / pixel-shuffle / importance map / quantiser kernels and the one-pass GDN (csrc/gdn_kernels.hip)."""
import torch
from torch import nn
import lic360
from lic360_operator import GDN, Dtow, SpherePad, SphereTrim, SphereCutEdge, QUANT, Dquant, SphereLatScaleNet, ImpMap


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride, pad)


FUSED_MIN_WORKGROUPS = 256          # lic360.sconv3x3 is used when its workgroups (330 us each) fill whole rounds of the 256 CUs to 80 % or more: 512 per
                                    # 260x516 map -- one image is two exact rounds -- but not 128 (measured against MIOpen, batches 1..8: tools/conv3x3_probe.py)


FUSED_MIN_FILL = 0.8


def _fusable(conv, x, ring, ring_w=None, mod=None):
    """does this 3x3 stride-1 convolution of a map with x's batch, height and width run on lic360.sconv3x3?  (inference only: the kernels have
    no backward -- with gradients enabled the fused path is taken only if neither x nor ANY parameter of the block `mod` that the path pushes
    through them (the 1x1 layers, shortcuts, PReLU slopes, biases: all of the block) requires one)"""
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or (mod is not None and any(p.requires_grad for p in mod.parameters()))):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and conv.bias is not None):
        return False
    cout, cin = conv.weight.shape[:2]
    if not lic360.sconv3x3_supported(cin, cout):
        return False
    n, _, hp, wp = x.shape
    nr, nc = hp - 2 * ring, wp - 2 * (ring if ring_w is None else ring_w)
    rows = nr // 16 if 0 < nr % 16 <= (2 if cout % 192 == 0 else 4) and nr >= 16 else (nr + 15) // 16      # (a short remainder rides on the last tile row)
    tiles = n * rows * ((nc + 15) // 16) * (cout // 192 if cout % 192 == 0 else 1)
    return tiles >= FUSED_MIN_WORKGROUPS and tiles >= FUSED_MIN_FILL * 256 * ((tiles + 255) // 256)


def _packed(conv):
    """the conv's weight in lic360.sconv3x3's / sconv1x1's operand order, repacked when the parameter was written (its version counter) or moved"""
    key = (conv.weight.data_ptr(), conv.weight._version)
    if getattr(conv, "_s3_key", None) != key:
        pack = lic360.sconv3x3_pack if conv.kernel_size == (3, 3) else lic360.sconv1x1_pack
        conv._s3_packed, conv._s3_key = pack(conv.weight.detach()), key
    return conv._s3_packed


def _scratch(shape, like):
    """an intermediate of a fused block: a fresh buffer per call from the caching allocator (which tracks the stream it is used on), so one model may
    run on several HIP streams at once (module-owned buffers raced there).  Its apron is NOT initialised and need not be: every consumer
    reads aprons by index from the interior (sconv3x3 with sphere != 0), refreshes them in place (SpherePad), works position by position and is trimmed
    afterwards (1x1 layers, PReLU, GDN), or overwrites them (SphereTrim, sphere_apron_from)."""
    return torch.empty(tuple(shape), dtype=torch.float32, device=like.device)


class ResidualBlock(nn.Module):
    """1x1 -> 3x3 -> 1x1 bottleneck on a refreshed apron (model_zoo.py:8-23)"""
    def __init__(self, channels, device_id=0):
        super().__init__()
        half = channels // 2
        self.pad = SpherePad(2, device_id, True)
        self.conv1, self.relu1 = _conv(channels, half, 1), nn.PReLU(half)
        self.conv2, self.relu2 = _conv(half, half, 3, 1, 1), nn.PReLU(half)
        self.conv3 = _conv(half, channels, 1)
        self.trim = SphereTrim(2, device_id)

    def forward(self, x):
        if _fusable(self.conv2, x, 2, mod=self) and x.is_contiguous():
            # conv1 and PReLU are pointwise, so the apron of relu1(conv1(pad(x))) is the sphere wrap of its own interior: conv2 reads it
            # by index and x needs no refresh; only the interior of conv2's output survives the final trim.  The 1x1 layers run on the same
            # kernel body with their PReLU / residual add in the epilogue (interior window only) when their shapes allow it.
            n, c, hp, wp = x.shape
            one = lic360.sconv1x1_supported(c, c // 2) and lic360.sconv1x1_supported(c // 2, c) and self.conv1.bias is not None and self.conv3.bias is not None
            if one:
                y = _scratch((n, c // 2, hp, wp), x)
                lic360.sconv1x1(x, _packed(self.conv1), self.conv1.bias, self.relu1.weight, None, y, ring=2)
            else:
                y = self.relu1(self.conv1(x)).contiguous()
            y2 = _scratch(y.shape, y)
            lic360.sconv3x3(y, _packed(self.conv2), self.conv2.bias, self.relu2.weight, None, y2, pad=2, sphere=True, ring=2)
            if one:
                out = torch.empty_like(x)
                lic360.sconv1x1(y2, _packed(self.conv3), self.conv3.bias, None, x, out, ring=2)
                return self.trim(out)
            return self.trim(x + self.conv3(y2))
        y = self.pad(x)
        y = self.relu2(self.conv2(self.relu1(self.conv1(y))))
        return self.trim(x + self.conv3(y))


class AttentionBlock(nn.Module):
    """x + trunk(x) * sigmoid-gate(x) (model_zoo.py:25-46)"""
    def __init__(self, channels, device_id=0):
        super().__init__()
        three = lambda: [ResidualBlock(channels, device_id) for _ in range(3)]
        self.trunk = nn.Sequential(*three())
        self.attention = nn.Sequential(*three(), _conv(channels, channels, 1), nn.Sigmoid())

    def forward(self, x):
        if _fusable(self.trunk[0].conv2, x, 2, mod=self.trunk[0]):
            x = self.trunk[0].pad(x)                                       # the reference's first ResidualBlock refreshes x's apron in place: `x + ...` below carries it
        return x + self.trunk(x) * self.attention(x)


class ResidualBlockV2(nn.Module):
    """two 3x3 convs whose receptive field is fed by the apron (model_zoo.py:48-64)"""
    def __init__(self, channels, device_id):
        super().__init__()
        self.pad = SpherePad(2, device_id, True)
        self.conv1, self.relu1, self.trim1 = _conv(channels, channels, 3, 1, 1), nn.PReLU(channels), SphereTrim(1, device_id)
        self.conv2, self.relu2, self.trim2 = _conv(channels, channels, 3, 1, 1), nn.PReLU(channels), SphereTrim(2, device_id)

    def forward(self, x):
        if _fusable(self.conv1, x, 1, 2, mod=self) and x.is_contiguous():
            # conv1 over the apron read by index, output on the rows of the 1-ring window (the outermost ring is never read) and on the
            # interior's columns only: its input is periodic in longitude, so its 1-ring COLUMNS would repeat its interior's last / first
            # column bit for bit -- conv2 reads them there (longitude wrap), reads the rows as they are, adds x on the interior; the
            # output's apron is x's refreshed apron, as `x + trim2(...)` leaves it in the reference
            y1 = _scratch(x.shape, x)
            lic360.sconv3x3(x, _packed(self.conv1), self.conv1.bias, self.relu1.weight, None, y1, pad=2, sphere=1, ring=1, ring_w=2)
            out = torch.empty_like(x)
            lic360.sconv3x3(y1, _packed(self.conv2), self.conv2.bias, self.relu2.weight, x, out, pad=2, sphere=2, ring=2)
            return lic360.sphere_apron_from(x, out, 2)
        y = self.trim1(self.relu1(self.conv1(self.pad(x))))
        return x + self.trim2(self.relu2(self.conv2(y)))


class ResidualBlockDown(nn.Module):
    """stride-2 stage with a GDN branch and a 1x1 stride-2 shortcut; `hidden=False` is the first stage, whose input has no apron yet
    (model_zoo.py:66-95)"""
    def __init__(self, channels, channel_in, device_id, hidden=True):
        super().__init__()
        self.pad1 = SpherePad(2, device_id, hidden)
        self.conv1, self.relu1 = _conv(channel_in, channels, 3, 2, 3), nn.PReLU(channels)
        self.trim = SphereTrim(2, device_id)
        self.pad2 = SpherePad(2, device_id, True)
        self.conv2, self.relu2 = _conv(channels, channels, 3, 1, 1), GDN(channels, device_id)
        self.short_cut = _conv(channel_in, channels, 1, 2, 2)
        self.hidden = hidden

    def forward(self, x):
        if self.hidden:
            skip = self.short_cut(x)                                        # before pad1 refreshes the apron in place
            y = self.pad1(x)
        else:
            x = self.pad1(x)
            skip, y = None, x
        # (the reference trims here and refreshes the same apron at once -- `trim` then `pad2` in place, model_zoo.py:83-84,90-91: the refresh
        #  overwrites every cell the trim zeroed, and the fused conv2 does not read the apron at all: one launch less on either path, same result)
        y = self.relu1(self.conv1(y))
        if torch.is_grad_enabled() and y.requires_grad:
            y = self.trim(y)                                                # a recording pass keeps the reference's sequence: the in-place pad's backward leaves the apron's gradient for the trim's to zero
        if _fusable(self.conv2, y, 2, mod=self) and y.is_contiguous():
            # conv2 reads the apron of y by index (no pad2); GDN is pointwise over positions, its frame cells are trimmed below
            y2 = _scratch(y.shape, y)
            y = self.relu2(lic360.sconv3x3(y, _packed(self.conv2), self.conv2.bias, None, None, y2, pad=2, sphere=True, ring=2))
        else:
            y = self.relu2(self.conv2(self.pad2(y)))
        return self.trim((self.short_cut(x) if skip is None else skip) + y)


class SphereConv2(nn.Module):
    def __init__(self, channel_in, channel_out, kernel_size, stride, pad=0, device_id=0):
        super().__init__()
        self.conv = _conv(channel_in, channel_out, kernel_size, stride, pad)
        self.pad, self.trim = SpherePad(2, device_id, True), SphereTrim(2, device_id)

    def forward(self, x):
        return self.trim(self.conv(self.pad(x)))


class SphereConv3(SphereConv2):
    """the same with an out-of-place pad: the synthesis side's first layer, whose input has no apron (model_zoo.py:158-168)"""
    def __init__(self, channel_in, channel_out, kernel_size, stride, pad=0, device_id=0):
        super().__init__(channel_in, channel_out, kernel_size, stride, pad, device_id)
        self.pad = SpherePad(2, device_id, False)


class EncoderV2(nn.Module):
    """image [n,3,512,1024] -> (code in (0,1) [n,cc,32,64], importance map [n,1,32,64]); every map carries a 2-cell sphere apron
    until the final cut (model_zoo.py:108-143)"""
    def __init__(self, channels, code_channels, device_id):
        super().__init__()
        d = device_id
        self.net = nn.Sequential(ResidualBlockDown(channels, 3, d, False), ResidualBlockV2(channels, d), ResidualBlockDown(channels, channels, d),
                                 AttentionBlock(channels, d), ResidualBlockV2(channels, d), ResidualBlockDown(channels, channels, d),
                                 ResidualBlockV2(channels, d), SphereConv2(channels, channels, 3, 2, 3, d))
        self.net2 = nn.Sequential(AttentionBlock(channels, d), _conv(channels, code_channels, 1), SphereCutEdge(2, d), nn.Sigmoid())
        self.imp_net = nn.Sequential(ResidualBlockV2(channels, d), ResidualBlockV2(channels, d), _conv(channels, 1, 1), nn.Sigmoid(),
                                     SphereCutEdge(2, d), SphereLatScaleNet(512 // 16, d))
        self.imp_net[2].bias.data.fill_(3)

    def forward(self, x):
        t = self.net(x)
        return self.net2(t), self.imp_net(t)


class ResidualBlockUp(nn.Module):
    """x2 upsampling stage: conv to 4c + pixel shuffle, an inverse-GDN branch, a 1x1 shortcut through its own shuffle (model_zoo.py:145-170)"""
    def __init__(self, channels, device_id):
        super().__init__()
        d = device_id
        self.pad1 = SpherePad(2, d, True)
        self.conv1, self.relu1 = _conv(channels, channels * 4, 3, 1), nn.PReLU(channels * 4)
        self.dtow1, self.trim1 = Dtow(2, True, d), SphereTrim(2, d)
        self.pad2 = SpherePad(2, d, True)
        self.conv2, self.relu2 = _conv(channels, channels, 3, 1, 1), GDN(channels, d, inverse=True)
        self.short_cut = _conv(channels, channels * 4, 1)
        self.cut_edge, self.dtow2, self.trim2 = SphereCutEdge(1, d), Dtow(2, True, d), SphereTrim(2, d)

    def forward(self, x):
        if _fusable(self.conv1, x, 2, mod=self) and x.is_contiguous():
            # the unpadded conv1 (output (h+2) x (w+2)) with PReLU in its epilogue, on the INTERIOR's window only: its 1-ring outputs become the
            # 2-ring of the shuffled map, which trim1 zeroes; x's apron is read by index (no pad1), and the shortcut's apron cells end in the
            # rings trim2 zeroes
            # ... and the pixel shuffle is the kernel's store pattern (a lane's four accumulator registers are one 2 x 2 block of the shuffled
            # map): conv1 -> PReLU -> Dtow -> trim1 in one launch.  The window lands exactly on the shuffled map's interior; its apron is
            # never read (conv2 reads aprons by index, or pad2 refreshes it), so trim1 has nothing to do.
            n, c, hp, wp = x.shape
            b = _scratch((n, self.conv1.out_channels // 4, 2 * (hp - 2), 2 * (wp - 2)), x)
            lic360.sconv3x3(x, _packed(self.conv1), self.conv1.bias, self.relu1.weight, None, b, pad=2, sphere=1, ring=2, crop=1, shuffle=True)
        else:
            b = self.dtow1(self.relu1(self.conv1(self.pad1(x))))           # (trim1 -> pad2 in place, model_zoo.py:160-161: the refresh overwrites what the trim zeroed)
            if torch.is_grad_enabled() and b.requires_grad:
                b = self.trim1(b)                                           # (recording pass: the reference's sequence, see ResidualBlockDown)
        if _fusable(self.conv2, b, 2, mod=self) and b.is_contiguous():
            b2 = _scratch(b.shape, b)
            b = self.relu2(lic360.sconv3x3(b, _packed(self.conv2), self.conv2.bias, None, None, b2, pad=2, sphere=True, ring=2))
        else:
            b = self.relu2(self.conv2(self.pad2(b)))
        c_in, c_out = self.short_cut.in_channels, self.short_cut.out_channels
        if (_fusable(self.conv2, b, 2, mod=self) and x.is_contiguous() and b.is_contiguous() and lic360.sconv1x1_supported(c_in, c_out) and self.short_cut.bias is not None
                and tuple(b.shape) == (x.shape[0], c_out // 4, 2 * (x.shape[2] - 2), 2 * (x.shape[3] - 2))):
            # the shortcut -- cut_edge(1) -> 1x1 conv to 4c -> Dtow(2) -- and the `b +` in one launch: the 1x1 instantiation with the shuffled store,
            # b as its (shuffled) residual, on the interior window (trim2 zeroes the rest)
            out = torch.empty_like(b)
            lic360.sconv1x1(x, _packed(self.short_cut), self.short_cut.bias, None, b, out, ring=2, crop=1, shuffle=True)
            return self.trim2(out)
        return self.trim2(b + self.dtow2(self.short_cut(self.cut_edge(x))))


class Decoder(nn.Module):
    """latent [n,cc,32,64] -> image [n,3,512,1024] (model_zoo.py:172-205)"""
    def __init__(self, channels, code_channels, device_id):
        super().__init__()
        d = device_id
        self.net = nn.Sequential(SphereConv3(code_channels, channels, 1, 1, 0, d), AttentionBlock(channels, d), ResidualBlockV2(channels, d),
                                 ResidualBlockUp(channels, d), ResidualBlockV2(channels, d), ResidualBlockUp(channels, d), AttentionBlock(channels, d),
                                 ResidualBlockV2(channels, d), ResidualBlockUp(channels, d), ResidualBlockV2(channels, d), SpherePad(2, d, True),
                                 _conv(channels, 12, 3, 1, 1), SphereCutEdge(2, d), Dtow(2, True, d))

    def forward(self, x):
        return self.net(x)


class CMP_Encoder(nn.Module):
    """image -> (symbols 0..7 under the mask [n,48,64,128], mask [n,48,64,128], importance levels [n,1,32,64]): what EntEncoderFast /
    FusedCodec.encode and ImpEntEncoderFast / FusedImpCodec.encode take (model_zoo.py:334-356)"""
    def __init__(self, channels=192, code_channels=192, quant_levels=8, gpu_id=0):
        super().__init__()
        self.encoder = EncoderV2(channels, code_channels, gpu_id)
        self.imp_level = code_channels // 4
        self.quant = QUANT(code_channels, quant_levels, device_id=gpu_id, ntop=2)
        self.imp = ImpMap(1, 0.0001, 0.0001, self.imp_level, 0.618, 0.618, 3, gpu_id, 2)
        self.dtw1, self.dtw2 = Dtow(2, True, gpu_id), Dtow(2, True, gpu_id)

    def forward(self, x):
        code, imap = self.encoder(x)
        tcode, mask, _ = self.imp(code, imap)
        _, qy = self.quant(tcode)
        return self.dtw1(qy), self.dtw2(mask), torch.sum(mask, dim=1, keepdim=True) / 4


class CMP_Decoder(nn.Module):
    """(symbols, mask) as decoded from the bitstreams -> image (model_zoo.py:358-379)"""
    def __init__(self, channels=192, code_channels=192, quant_levels=8, gpu_id=0):
        super().__init__()
        self.decoder = Decoder(channels, code_channels, gpu_id)
        self.imp_level = code_channels // 4
        self.quant = Dquant(code_channels, quant_levels, device=gpu_id)
        self.dtw1, self.dtw2 = Dtow(2, False, gpu_id), Dtow(2, False, gpu_id)

    def forward(self, code, mask):
        return self.decoder(self.quant(self.dtw1(code), self.dtw2(mask)))


def transform_gflops(channels=192, code_channels=192, h=512, w=1024):
    """nominal conv + GDN GFLOP of one analysis and one synthesis pass (2 flops per MAC; aprons counted: every map is 4 cells larger)"""
    def conv(cin, cout, k, hh, ww):
        return 2.0 * cin * cout * k * k * hh * ww
    c, enc, dec = channels, 0.0, 0.0
    bott = lambda hh, ww: conv(c, c // 2, 1, hh, ww) + conv(c // 2, c // 2, 3, hh, ww) + conv(c // 2, c, 1, hh, ww)
    att = lambda hh, ww: 6 * bott(hh, ww) + conv(c, c, 1, hh, ww)
    v2 = lambda hh, ww: 2 * conv(c, c, 3, hh, ww)
    hh, ww, cin = h, w, 3
    for stage in range(3):                                                  # three down stages
        hh, ww = hh // 2 + 4, ww // 2 + 4
        enc += conv(cin, c, 3, hh, ww) + conv(c, c, 3, hh, ww) + conv(c, c, 1, hh, ww) + conv(cin, c, 1, hh, ww) + v2(hh, ww)
        if stage == 1:
            enc += att(hh, ww)
        cin = c
        hh, ww = hh - 4, ww - 4
    hh, ww = hh // 2 + 4, ww // 2 + 4                                       # 36 x 68
    enc += conv(c, c, 3, hh, ww) + att(hh, ww) + conv(c, code_channels, 1, hh, ww) + 2 * v2(hh, ww) + conv(c, 1, 1, hh, ww)
    dec += conv(code_channels, c, 1, hh, ww) + att(hh, ww) + v2(hh, ww)
    for stage in range(3):                                                  # three up stages
        dec += conv(c, 4 * c, 3, hh, ww) + conv(c, 4 * c, 1, hh, ww)
        hh, ww = (hh - 4) * 2 + 4, (ww - 4) * 2 + 4
        dec += conv(c, c, 3, hh, ww) + conv(c, c, 1, hh, ww) + v2(hh, ww)
        if stage == 1:
            dec += att(hh, ww)
    dec += conv(c, 12, 3, hh, ww)
    return enc / 1e9, dec / 1e9
