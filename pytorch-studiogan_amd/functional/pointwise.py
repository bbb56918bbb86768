"""small elementwise operators of the residual blocks (pooling, add, ReLU and their masks)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)
from .conv import _GRAD_LINK

# ---------------------------------------------------------------------------------------------------------
# small elementwise ops of the D blocks
# ---------------------------------------------------------------------------------------------------------
class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.call("sg_avgpool2_fwd", L.dt(x), L.ptr(x), L.ptr(y), N, H, W, Cc, L.stream())
        ctx.shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():
            return AvgPool2BwdFn.apply(dy, ctx.shape)
        return _avgpool2_bwd(dy, ctx.shape)


def _avgpool2_bwd(dy, shape):
    dy = _c(dy)
    N, H, W, Cc = shape
    dx = torch.empty((N, H, W, Cc), dtype=dy.dtype, device=dy.device)
    L.call("sg_avgpool2_bwd", L.dt(dy), L.ptr(dy), L.ptr(dx), N, H, W, Cc, L.stream())
    return dx


class AvgPool2BwdFn(torch.autograd.Function):
    """0.25 * broadcast of the pooled gradient; its adjoint is the pooling itself (second-order pass)."""

    @staticmethod
    def forward(ctx, dy, shape):
        ctx.shape = shape
        return _avgpool2_bwd(dy, shape)

    @staticmethod
    def backward(ctx, ddx):
        ddx = _c(ddx)
        N, H, W, Cc = ctx.shape
        g = torch.empty((N, H // 2, W // 2, Cc), dtype=ddx.dtype, device=ddx.device)
        L.call("sg_avgpool2_fwd", L.dt(ddx), L.ptr(ddx), L.ptr(g), N, H, W, Cc, L.stream())
        return g, None


class AddFn(torch.autograd.Function):
    """out = a + x (identity skip of a BN DiscBlock)."""

    @staticmethod
    def forward(ctx, a, x):
        a, x = _c(a), _c(x)
        out = torch.empty_like(a)
        L.call("sg_convert", L.dt(a), L.dt(a), L.ptr(a), L.ptr(out), a.numel(), L.stream())
        L.call("sg_axpby", L.dt(a), L.ptr(x), L.ptr(out), a.numel(), 1.0, 1.0, L.stream())
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def _mask(dy, x):
    dy = _c(dy)
    dx = torch.empty_like(x)
    L.call("sg_relu_mask", L.dt(x), L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), L.stream())
    return dx


class MaskFn(torch.autograd.Function):
    """dy * (x > 0) as a differentiable op of dy (x's mask is piecewise constant): the ReLU backward inside a create_graph pass."""

    @staticmethod
    def forward(ctx, dy, x):
        ctx.save_for_backward(x)
        return _mask(dy, x)

    @staticmethod
    def backward(ctx, dd):
        (x,) = ctx.saved_tensors
        return _mask(dd, x), None


class ReluFn(torch.autograd.Function):
    """standalone ReLU (only where no neighbouring launch can absorb it): y = x * (x > 0). link: a GradLink shared with another reader of the same x whose
    backward runs later (the first convolution of a BigGAN-deep discriminator block): the masked gradient is stashed there and rides as the residual of that
    convolution's data-gradient launch."""

    @staticmethod
    def forward(ctx, x, link=None):
        x = _c(x)
        y = torch.empty_like(x)
        L.call("sg_relu_mask", L.dt(x), L.ptr(x), L.ptr(x), L.ptr(y), x.numel(), L.stream())
        ctx.save_for_backward(x)
        ctx.link = link
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return MaskFn.apply(dy, x), None
        dx = _mask(dy, x)
        if ctx.link is not None and _GRAD_LINK[0]:
            ctx.link.dx, dx = dx, None
        return dx, None


class AddReluFn(torch.autograd.Function):
    """out = a + relu(x): the identity-skip DiscBlock (its in-place ReLU also rewrites the skip tensor;
    reference src/models/big_resnet.py:221-242 with nn.ReLU(inplace=True), src/config.py:476)."""

    @staticmethod
    def forward(ctx, a, x):
        a, x = _c(a), _c(x)
        out = torch.empty_like(a)
        L.call("sg_add_relu", L.dt(a), L.ptr(a), L.ptr(x), L.ptr(out), a.numel(), L.stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[1]:
            dx = MaskFn.apply(dy, x) if torch.is_grad_enabled() else _mask(dy, x)
        return dy, dx


__all__ = ['AddFn', 'AddReluFn', 'AvgPool2BwdFn', 'AvgPool2Fn', 'MaskFn', 'ReluFn', '_avgpool2_bwd', '_mask']
