"""bias_act: y = clamp(gain * act(x + b)) with first and second order gradients (reference src/utils/style_ops/bias_act.py:61-205; kernel
contract bias_act.cu:23-147 -> csrc/style.hip sg_bias_act). The activation table (names, default alpha / gain, which tensors the gradient
needs, whether a second derivative exists) is the reference's `activation_funcs` (bias_act.py:20-30)."""
import math

import torch

from .. import _lib as L


class _Spec:
    def __init__(self, idx, def_alpha, def_gain, ref, has_2nd_grad):
        self.cuda_idx, self.def_alpha, self.def_gain, self.ref, self.has_2nd_grad = idx, def_alpha, def_gain, ref, has_2nd_grad


SQRT2 = math.sqrt(2.0)
activation_funcs = {
    "linear": _Spec(1, 0.0, 1.0, "", False), "relu": _Spec(2, 0.0, SQRT2, "y", False), "lrelu": _Spec(3, 0.2, SQRT2, "y", False),
    "tanh": _Spec(4, 0.0, 1.0, "y", True), "sigmoid": _Spec(5, 0.0, 1.0, "y", True), "elu": _Spec(6, 0.0, 1.0, "y", True),
    "selu": _Spec(7, 0.0, 1.0, "y", True), "softplus": _Spec(8, 0.0, 1.0, "y", True), "swish": _Spec(9, 0.0, SQRT2, "x", True),
}


def _launch(x, b, xref, yref, dy, grad, dim, spec, alpha, gain, clamp):
    """one sg_bias_act launch on dense tensors of one dtype; returns the output (same shape / memory format as x)"""
    if not x.is_cuda:
        raise RuntimeError("bias_act: the HIP kernels need a GPU tensor (no CPU fallback on the product path)")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"bias_act: float32 / bfloat16 only, got {x.dtype}")
    y = torch.empty_like(x)
    n = x.numel()
    if n == 0:
        return y
    step_b, size_b = 1, 1
    if b is not None:
        if b.dim() != 1 or b.shape[0] != x.shape[dim]:
            raise RuntimeError("bias_act: b must be a vector matching x.shape[dim]")       # bias_act.cpp:45-47
        # stride of `dim` in the dense layout actually in memory (contiguous, or channels_last where dim 1 has stride 1)
        step_b, size_b = x.stride(dim), b.shape[0]
        b = b.to(x.dtype).contiguous()
    for t in (xref, yref, dy):
        if t is not None and (t.shape != x.shape or t.dtype != x.dtype or t.stride() != x.stride()):
            raise RuntimeError("bias_act: xref / yref / dy must have the shape, dtype and layout of x")   # bias_act.cpp:40-44
    L.call("sg_bias_act", L.dt(x), L.ptr(x), L.ptr(b), L.ptr(xref), L.ptr(yref), L.ptr(dy), L.ptr(y), n, step_b, size_b, grad, spec.cuda_idx,
           float(alpha), float(gain), float(clamp), L.stream())
    return y


def _dense(t, fmt):
    return t.contiguous(memory_format=fmt)


_cache = {}


def _make(dim, act, alpha, gain, clamp):
    key = (dim, act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    spec = activation_funcs[act]

    class BiasAct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            ctx.fmt = torch.channels_last if x.dim() == 4 and x.stride(1) == 1 and x.shape[1] > 1 else torch.contiguous_format
            x = _dense(x, ctx.fmt)
            y = x
            if act != "linear" or gain != 1 or clamp >= 0 or b is not None:
                y = _launch(x, b, None, None, None, 0, dim, spec, alpha, gain, clamp)
            keep_x = "x" in spec.ref or spec.has_2nd_grad
            ctx.save_for_backward(x if keep_x else None, b if (keep_x and b is not None) else None, y if "y" in spec.ref else None)
            ctx.has_b = b is not None
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = _dense(dy, ctx.fmt)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or (ctx.has_b and ctx.needs_input_grad[1]):
                dx = dy
                if act != "linear" or gain != 1 or clamp >= 0:
                    dx = BiasActGrad.apply(dy, x, b, y)
            if ctx.has_b and ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.dim()) if i != dim])
            return dx, db

    class BiasActGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.fmt = torch.channels_last if dy.dim() == 4 and dy.stride(1) == 1 and dy.shape[1] > 1 else torch.contiguous_format
            dx = _launch(dy, b, x, y, None, 1, dim, spec, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = _dense(d_dx, ctx.fmt)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _launch(d_dx, b, x, y, dy, 2, dim, spec, alpha, gain, clamp)
            if spec.has_2nd_grad and b is not None and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.dim()) if i != dim])
            return d_dy, d_x, d_b, None

    _cache[key] = BiasAct
    return BiasAct


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="cuda"):
    """Adds bias `b` along dimension `dim`, evaluates `act`, scales by `gain`, clamps to [-clamp, clamp]; every step optional
    (reference bias_act.py:61-85). First and second order gradients; no third order."""
    if not isinstance(x, torch.Tensor):
        raise AssertionError("bias_act: x must be a tensor")
    if impl not in ("ref", "cuda"):
        raise AssertionError("bias_act: impl must be 'ref' or 'cuda'")
    if act not in activation_funcs:
        raise KeyError(act)
    if clamp is not None and clamp < 0:
        raise AssertionError("bias_act: clamp must be >= 0")
    if b is not None and not (0 <= dim < x.dim()):
        raise AssertionError("bias_act: dim out of range")
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    return _make(dim, act, alpha, gain, clamp).apply(x, b)
