"""layout conversions at the reference's NCHW fp32 boundary (csrc/elementwise.hip)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

# ---------------------------------------------------------------------------------------------------------
# layout at the reference's NCHW fp32 boundary
# ---------------------------------------------------------------------------------------------------------
class NchwToNhwcFn(torch.autograd.Function):
    """fp32 NCHW -> compute-dtype NHWC (D input image; G's linear0 output). cpad > C: the NHWC tensor gets cpad channels, the extra
    ones zero (RGB images travel as 8-channel tensors so every convolution uses the 16-byte loaders)."""

    @staticmethod
    def forward(ctx, x, dtype, cpad=0):
        x = _c(x)
        N, Cc, H, W = x.shape
        ld = max(cpad, Cc)
        y = torch.empty((N, H, W, ld), dtype=dtype, device=x.device)      # (sg_nchw_to_nhwc writes the zero padding of ld > C rows itself)
        L.call("sg_nchw_to_nhwc", L.dt(dtype), L.ptr(x), L.ptr(y), N, Cc, H, W, ld, L.stream())
        ctx.channels = Cc
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():        # create_graph=True (WGAN-GP): stay on differentiable ops
            return NhwcToNchwFn.apply(dy, False, ctx.channels), None, None
        dy = _c(dy)
        N, H, W, ld = dy.shape
        Cc = ctx.channels
        dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dy.device)
        L.call("sg_nhwc_to_nchw", L.dt(dy), L.ptr(dy), L.ptr(dx), N, Cc, H, W, ld, 0, L.stream())
        return dx, None, None


class NhwcToNchwFn(torch.autograd.Function):
    """compute-dtype NHWC -> fp32 NCHW with optional tanh (G output image). channels < x.shape[3]: only the first `channels`
    are real (the last convolution of G writes 8-channel rows)."""

    @staticmethod
    def forward(ctx, x, apply_tanh, channels=0):
        x = _c(x)
        N, H, W, ld = x.shape
        Cc = channels or ld
        y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
        L.call("sg_nhwc_to_nchw", L.dt(x), L.ptr(x), L.ptr(y), N, Cc, H, W, ld, 1 if apply_tanh else 0, L.stream())
        ctx.apply_tanh = apply_tanh
        ctx.in_dtype = x.dtype
        ctx.ld = ld
        if apply_tanh:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():        # create_graph=True (latent optimisation: d D(G(z)) / dz is differentiated again): stay on differentiable operators
            t = TanhGradFn.apply(dy, ctx.saved_tensors[0]) if ctx.apply_tanh else dy
            return NchwToNhwcFn.apply(t, ctx.in_dtype, ctx.ld), None, None
        dy = _c(dy.float())
        N, Cc, H, W = dy.shape
        y = ctx.saved_tensors[0] if ctx.apply_tanh else None
        ld = ctx.ld
        dx = (torch.zeros if ld > Cc else torch.empty)((N, H, W, ld), dtype=ctx.in_dtype, device=dy.device)
        L.call("sg_nchw_grad_to_nhwc", L.dt(ctx.in_dtype), L.ptr(dy), L.ptr(y), L.ptr(dx), N, Cc, H, W, ld, 1 if ctx.apply_tanh else 0, L.stream())
        return dx, None, None


class TanhGradFn(torch.autograd.Function):
    """t = dy * (1 - y^2) with y = tanh(x) the forward's OUTPUT (so its gradient re-enters the producing node): the tanh backward as a differentiable operator"""

    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _c(dy.float()), _c(y.float())
        t = torch.empty_like(dy)
        L.call("sg_tanh_bwd", L.ptr(dy), L.ptr(y), L.ptr(t), dy.numel(), L.stream())
        ctx.save_for_backward(dy, y)
        return t

    @staticmethod
    def backward(ctx, g):
        _first_order_only("TanhGradFn")
        dy, y = ctx.saved_tensors
        g = _c(g.float())
        g_dy = g_y = None
        if ctx.needs_input_grad[0]:
            g_dy = torch.empty_like(g)
            L.call("sg_tanh_bwd", L.ptr(g), L.ptr(y), L.ptr(g_dy), g.numel(), L.stream())
        if ctx.needs_input_grad[1]:
            g_y = torch.empty_like(g)
            L.call("sg_tanh_bwd2", L.ptr(g), L.ptr(dy), L.ptr(y), L.ptr(g_y), g.numel(), L.stream())
        return g_dy, g_y


class ConvertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        x = _c(x)
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        L.call("sg_convert", L.dt(x), L.dt(dtype), L.ptr(x), L.ptr(y), x.numel(), L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():          # a create_graph pass: the conversion is linear, its adjoint is the conversion back
            return ConvertFn.apply(dy, ctx.src), None
        dy = _c(dy)
        dx = torch.empty(dy.shape, dtype=ctx.src, device=dy.device)
        L.call("sg_convert", L.dt(dy), L.dt(ctx.src), L.ptr(dy), L.ptr(dx), dy.numel(), L.stream())
        return dx, None


__all__ = ['ConvertFn', 'NchwToNhwcFn', 'NhwcToNchwFn', 'TanhGradFn']
