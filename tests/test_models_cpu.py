"""lic360_models on a host without a GPU: the networks construct (native op objects are created lazily-launched), the parameter tree is the
reference's (test/model_zoo.py: checkpoints load by name), and the nominal flop count used by the bench is stable."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "360-image-compression_amd"))


def test_parameter_tree_and_flops():
    import lic360_models as lm
    enc, dec = lm.CMP_Encoder(32, 32, 8, 0), lm.CMP_Decoder(32, 32, 8, 0)
    ek, dk = enc.state_dict(), dec.state_dict()
    for k, shape in (("encoder.net.0.conv1.weight", (32, 3, 3, 3)), ("encoder.net.0.relu2.gamma", (32, 32)), ("encoder.net.0.short_cut.weight", (32, 3, 1, 1)),
                     ("encoder.net.1.relu1.weight", (32,)), ("encoder.net.3.trunk.2.conv3.weight", (32, 16, 1, 1)), ("encoder.net.3.attention.3.weight", (32, 32, 1, 1)),
                     ("encoder.net.7.conv.weight", (32, 32, 3, 3)), ("encoder.net2.1.weight", (32, 32, 1, 1)), ("encoder.imp_net.2.weight", (1, 32, 1, 1)),
                     ("encoder.imp_net.5.data", (1, 1, 32)), ("quant.weight", (32, 8)), ("quant.count", (32, 8))):
        assert tuple(ek[k].shape) == shape, k
    for k, shape in (("decoder.net.0.conv.weight", (32, 32, 1, 1)), ("decoder.net.3.conv1.weight", (128, 32, 3, 3)), ("decoder.net.3.relu2.beta", (32,)),
                     ("decoder.net.3.short_cut.weight", (128, 32, 1, 1)), ("decoder.net.11.weight", (12, 32, 3, 3)), ("quant.weight", (32, 8))):
        assert tuple(dk[k].shape) == shape, k
    assert float(ek["encoder.imp_net.2.bias"]) == 3.0                       # the importance head starts open (model_zoo.py:137)
    ge, gd = lm.transform_gflops()
    assert 440 < ge < 490 and 540 < gd < 590                                 # ~1 TFLOP per image and direction pair
    # full width: 60 / 57 convolutions as SURVEY.md 8f counts them
    import torch
    full_e, full_d = lm.EncoderV2(192, 192, 0), lm.Decoder(192, 192, 0)
    assert sum(isinstance(m, torch.nn.Conv2d) for m in full_e.modules()) == 60
    assert sum(isinstance(m, torch.nn.Conv2d) for m in full_d.modules()) == 57


def test_inplace_sphere_ops_do_not_break_backward():
    """UnaryFn used to mark_dirty() the in-place apron refresh / trim, which bumps x's version counter; a block that
    pads the same x in two branches after a conv has saved it (AttentionBlock, ResidualBlockDown(hidden)) then failed in
    backward with 'modified by an inplace operation'.  Mock op on CPU (the native op needs a GPU): the refresh is idempotent."""
    import os
    import sys
    import types
    import torch
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.modules.setdefault("lic360", types.ModuleType("lic360"))            # autograd.py itself imports nothing native
    sys.path.insert(0, os.path.join(here, "360-image-compression_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("lic360_autograd_cpu", os.path.join(here, "360-image-compression_amd", "lic360_operator", "autograd.py"))
    ag = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ag)

    class RefreshOp:                                     # in-place longitude wrap of a 1-px apron; backward = identity on this toy
        def forward(self, x):
            v = x.detach().numpy()                        # raw memory write, like the native op: no version-counter bump
            v[..., 0] = v[..., -2]
            v[..., -1] = v[..., 1]
            return [x]

        def backward(self, g):
            return [g]

    torch.manual_seed(0)
    conv1, conv2, conv3 = (torch.nn.Conv2d(2, 2, 3, padding=1) for _ in range(3))
    inp = torch.randn(1, 2, 6, 8, requires_grad=True)
    x = conv1(inp)
    op = RefreshOp()
    x = ag.UnaryFn.apply(x, op, True)                     # the block's first pad
    a = conv2(x)                                         # saves x
    xb = ag.UnaryFn.apply(x, op, True)                    # the second branch pads the same x again, in place
    b = conv3(xb)
    (a * b).sum().backward()                             # raised before the fix
    assert inp.grad is not None and torch.isfinite(inp.grad).all()
