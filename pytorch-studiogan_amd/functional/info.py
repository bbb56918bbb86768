"""InfoGAN's Q-head operators (csrc/ext/losses.hip)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

# ---------------------------------------------------------------------------------------------------------
# InfoGAN's Q heads (csrc/ext/losses.hip)
# ---------------------------------------------------------------------------------------------------------
class ExpFn(torch.autograd.Function):
    """y = exp(x) in fp32 (the continuous code's variance head, reference src/models/big_resnet.py:377)"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x.float())
        y = torch.empty_like(x)
        L.call("sg_exp_fwd", L.ptr(x), L.ptr(y), x.numel(), L.stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("ExpFn")
        (y,) = ctx.saved_tensors
        dy = _c(dy.float())
        dx = torch.empty_like(y)
        L.call("sg_exp_bwd", L.ptr(dy), L.ptr(y), L.ptr(dx), y.numel(), L.stream())
        return dx


class NormalNllFn(torch.autograd.Function):
    """reference src/utils/losses.py:369-375 normal_nll_loss(x, mu, var): value and the gradients w.r.t. mu and var from one launch (x: the sampled code)"""

    @staticmethod
    def forward(ctx, x, mu, var):
        x, mu, var = _c(x.detach().float()), _c(mu.float()), _c(var.float())
        if x.shape != mu.shape or mu.shape != var.shape or x.dim() != 2:
            raise RuntimeError("normal_nll_loss: three [B, K] tensors expected")
        loss = torch.empty(1, dtype=torch.float32, device=mu.device)
        dmu, dvar = torch.empty_like(mu), torch.empty_like(var)
        L.call("sg_normal_nll", L.ptr(x), L.ptr(mu), L.ptr(var), x.shape[0], x.shape[1], L.ptr(loss), L.ptr(dmu), L.ptr(dvar), L.stream())
        ctx.save_for_backward(dmu, dvar)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dmu, dvar = ctx.saved_tensors
        return None, dmu * g, dvar * g


__all__ = ['ExpFn', 'NormalNllFn']
