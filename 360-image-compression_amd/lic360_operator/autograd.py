"""torch.autograd glue for the native ops that have a gradient: the modules of this package call the plain,
no-grad path for inference and one of these functions when a gradient is being recorded.  Buffer semantics are the native ops'
(outputs and gradients are op-owned and reused by the next call of the same op), as with the reference's own wrappers."""
import torch


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def recording(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class UnaryFn(torch.autograd.Function):
    """y = op.forward(x)[0],  dL/dx = op.backward(dL/dy)[0]; `inplace`: the op overwrites x and returns it.
    The in-place ops (apron refresh of SpherePad, SphereTrim) are NOT marked dirty, exactly as the reference's wrappers
    (lic360_operator/SpherePad.py:9-15 returns the op's output, which is its input): autograd then aliases the result to x
    and x's version counter stays put, so a convolution that saved x before the refresh can still run its backward
    (AttentionBlock pads the same x in two branches, ResidualBlockDown pads after the shortcut has saved x).  That is sound
    because the refresh is idempotent on everything an earlier consumer read: it rewrites apron cells with the values a
    previous pad of the same interior already put there, and the trim's zeros are what the reference's graph sees too."""
    @staticmethod
    def forward(ctx, x, op, inplace):
        ctx.op = op
        return op.forward(x)[0]

    @staticmethod
    def backward(ctx, g):
        return ctx.op.backward(_c(g))[0], None, None


class LatScaleFn(torch.autograd.Function):
    """y[n,c,h,w] = x * weight[part(h)]; the band weights get the sum of g * x over their rows"""
    @staticmethod
    def forward(ctx, x, weight, op):
        ctx.op = op
        ctx.save_for_backward(x, weight)
        return op.forward(x, weight)[0]

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        gx = ctx.op.backward(g, weight)[0]
        npart = weight.numel()
        per_row = (g * x).sum((0, 1, 3))                                    # [h]
        return gx, per_row.view(npart, -1).sum(1).view_as(weight), None


class ImpMapFn(torch.autograd.Function):
    """(x, imp) -> (masked x[, mask], mean level); backward: the op's rule for the importance map, driven by the gap between the
    row-wise mean level and the op's latitude constraint (reference lic360_operator/ImpMap.py:8-57)"""
    @staticmethod
    def forward(ctx, x, imp, level, op, want_mask):
        imp = _c(torch.floor(imp * level) / level)
        out = op.forward(x, imp)
        ctx.op = op
        ctx.save_for_backward(imp, out[1])
        rt = torch.mean(imp)
        if want_mask:
            ctx.mark_non_differentiable(out[2])
            return out[0], out[2], rt
        return out[0], rt

    @staticmethod
    def backward(ctx, g, *unused):
        imp, constrain = ctx.saved_tensors
        gap = _c(torch.mean(imp, dim=3) - constrain)
        gx, gimp = ctx.op.backward(_c(g), imp, gap)
        return gx, gimp, None, None, None


class QuantFn(torch.autograd.Function):
    """(x, weight, count) -> quantised x[, index]; gradients: straight through for x, the op's level-increment gradient for weight,
    and the op's level counts as the "gradient" of count (so that an optimiser step accumulates them; reference QUANT.py:7-28)"""
    @staticmethod
    def forward(ctx, x, weight, count, op, training):
        out = op.forward(x, weight, count, training)
        ctx.op = op
        ctx.save_for_backward(x, out[0])
        return out[0] if len(out) == 1 else (out[0], out[1])

    @staticmethod
    def backward(ctx, *grads):
        x, y = ctx.saved_tensors
        gx, gw, cnt = ctx.op.backward([_c(g) for g in grads], x, y)
        return gx, gw, cnt.clone().detach(), None, None
