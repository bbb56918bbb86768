"""filtered_lrelu: bias -> upsample through fu -> leaky ReLU (gain, clamp) -> downsample through fd (StyleGAN3's alias-free non-linearity,
reference src/utils/style_ops/filtered_lrelu.py:54-155). The reference ships a 1284-line tiled CUDA kernel for it and, for shapes that kernel
does not take, runs exactly this chain of its own bias_act / upfirdn2d operators (filtered_lrelu.py:140-146, the `_filtered_lrelu_ref` path);
here the chain IS the implementation: four launches of csrc/style.hip kernels, each differentiable, so gradients of any order come from the
operators' own autograd Functions. (A single tiled launch that keeps the upsampled intermediate in LDS is the obvious next step; it changes
no result.)"""
import numpy as np
import torch

from . import bias_act as _bias_act
from . import upfirdn2d as _upfirdn2d


def _get_filter_size(f):
    if f is None:
        return 1, 1
    if not (isinstance(f, torch.Tensor) and 1 <= f.dim() <= 2):
        raise AssertionError("filter must be a 1-D or 2-D tensor")
    return int(f.shape[-1]), int(f.shape[0])     # width, height


def _parse_padding(padding):
    if isinstance(padding, (int, np.integer)):
        padding = [padding, padding]
    if not (isinstance(padding, (list, tuple)) and all(isinstance(v, (int, np.integer)) for v in padding)):
        raise AssertionError("padding must be an int or a list of ints")
    padding = [int(v) for v in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    if len(padding) != 4:
        raise AssertionError("padding must have 1, 2 or 4 entries")
    return tuple(padding)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False, impl="cuda"):
    """x: [N, C, H, W]; fu / fd: float32 FIR filters (1-D separable, 2-D, or None); b: per-channel bias of x's dtype; up / down: integer
    factors; padding relative to the upsampled image (negative = crop). Output [N, C, out_h, out_w] with
    out = (in * up + pad0 + pad1 - (fu - 1) - (fd - 1) + (down - 1)) // down per axis (reference filtered_lrelu.py:131-132)."""
    if not (isinstance(x, torch.Tensor) and x.dim() == 4):
        raise AssertionError("filtered_lrelu: x must be a 4-D NCHW tensor")
    if impl not in ("ref", "cuda"):
        raise AssertionError("filtered_lrelu: impl must be 'ref' or 'cuda'")
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    if b is not None:
        if not (isinstance(b, torch.Tensor) and b.dtype == x.dtype and b.dim() == 1 and b.shape[0] == x.shape[1]):
            raise AssertionError("filtered_lrelu: b must be a vector of x's dtype with one entry per channel")
    if not (isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1):
        raise AssertionError("filtered_lrelu: up / down must be integers >= 1")
    px0, px1, py0, py1 = _parse_padding(padding)
    if not (gain == float(gain) and gain > 0 and slope == float(slope) and slope >= 0):
        raise AssertionError("filtered_lrelu: gain must be > 0 and slope >= 0")
    if clamp is not None and not (clamp == float(clamp) and clamp >= 0):
        raise AssertionError("filtered_lrelu: clamp must be >= 0")
    N, C, in_h, in_w = x.shape
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    y = _bias_act.bias_act(x=x, b=b)                                                                                          # 1. bias
    y = _upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)           # 2.-4. upsample
    y = _bias_act.bias_act(x=y, act="lrelu", alpha=slope, gain=gain, clamp=clamp)                                             # 5.-7. leaky ReLU
    y = _upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)                                                   # 8.-9. downsample
    if tuple(y.shape) != (N, C, out_h, out_w) or y.dtype != x.dtype:
        raise AssertionError(f"filtered_lrelu: unexpected output {tuple(y.shape)} {y.dtype}")
    return y
