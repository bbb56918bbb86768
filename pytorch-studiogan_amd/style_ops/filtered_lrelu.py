"""filtered_lrelu: bias -> upsample through fu -> leaky ReLU (gain, clamp) -> downsample through fd (StyleGAN3's alias-free non-linearity,
reference src/utils/style_ops/filtered_lrelu.py:54-155). The reference ships a tiled CUDA kernel for it and, for what that kernel does not
take, the chain of its own bias_act / upfirdn2d operators (filtered_lrelu.py:140-146). Here:
  * forward with separable (1-D or absent) filters: ONE launch, csrc/style.hip sg_filtered_lrelu -- the up-sampled intermediate (up^2 times the
    input) stays in LDS instead of crossing HBM twice;
  * 2-D filters, tiles beyond the LDS budget, and every BACKWARD: the chain of sg_bias_act / sg_upfirdn2d launches, each a differentiable
    autograd Function, so gradients of any order come from the operators themselves (the backward re-runs the chain on the saved input:
    the fused forward keeps no intermediate)."""
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
    cfg = (up, down, (px0, px1, py0, py1), float(gain), float(slope), None if clamp is None else float(clamp), bool(flip_filter))
    separable = (fu is None or fu.dim() == 1) and (fd is None or fd.dim() == 1)
    if separable and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and _FUSED[0]:
        y = _FilteredLRelu.apply(x, b, fu, fd, cfg)
    else:
        y = _chain(x, b, fu, fd, cfg)
    if tuple(y.shape) != (N, C, out_h, out_w) or y.dtype != x.dtype:
        raise AssertionError(f"filtered_lrelu: unexpected output {tuple(y.shape)} {y.dtype}")
    return y


_FUSED = [True]        # tests / A-B: filtered_lrelu._FUSED[0] = False runs the operator chain for the forward as well


def _chain(x, b, fu, fd, cfg):
    up, down, (px0, px1, py0, py1), gain, slope, clamp, flip = cfg
    y = _bias_act.bias_act(x=x, b=b)                                                                                  # 1. bias
    y = _upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip)          # 2.-4. upsample
    y = _bias_act.bias_act(x=y, act="lrelu", alpha=slope, gain=gain, clamp=clamp)                                     # 5.-7. leaky ReLU
    return _upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip)                                               # 8.-9. downsample


class _FilteredLRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, fu, fd, cfg):
        from .. import _lib as L
        up, down, (px0, px1, py0, py1), gain, slope, clamp, flip = cfg
        xc = x.contiguous()
        N, C, H, W = xc.shape
        one = torch.ones(1, dtype=torch.float32, device=x.device)
        fu_d = one if fu is None else fu.to(device=x.device, dtype=torch.float32).contiguous()
        fd_d = one if fd is None else fd.to(device=x.device, dtype=torch.float32).contiguous()
        Wo = (W * up + px0 + px1 - (fu_d.numel() - 1) - (fd_d.numel() - 1) + (down - 1)) // down
        Ho = (H * up + py0 + py1 - (fu_d.numel() - 1) - (fd_d.numel() - 1) + (down - 1)) // down
        if Wo < 1 or Ho < 1:
            raise RuntimeError("filtered_lrelu: empty output")
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device)
        bc = None if b is None else b.to(x.dtype).contiguous()
        rc = 0
        if N * C > 0:
            rc = L.lib().sg_filtered_lrelu(L.dt(xc), L.ptr(xc), L.ptr(fu_d), L.ptr(fd_d), L.ptr(bc), L.ptr(y), N, C, H, W, fu_d.numel(), fd_d.numel(),
                                           up, down, px0, px1, py0, py1, gain, slope, -1.0 if clamp is None else clamp, 1 if flip else 0, L.stream())
        if rc == -3:                    # tile beyond the LDS budget: the chain has no such limit
            with torch.no_grad():
                y = _chain(x, b, fu, fd, cfg)
        elif rc != 0:
            raise RuntimeError("sg_filtered_lrelu: " + L.lib().sg_last_error().decode())
        ctx.save_for_backward(x, b, fu, fd)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, fu, fd = ctx.saved_tensors
        need_x, need_b = ctx.needs_input_grad[0], (b is not None and ctx.needs_input_grad[1])
        ctx.second_order = torch.is_grad_enabled()
        if not (need_x or need_b):
            return None, None, None, None, None
        # re-run the differentiable operator chain on the saved input and differentiate it (the fused forward keeps no intermediate);
        # under create_graph the result stays differentiable through the operators' own backward Functions
        with torch.enable_grad():
            second_order = ctx.second_order
            xr = x if second_order else x.detach().requires_grad_(need_x)         # create_graph: stay connected to the caller's graph
            br = b if (second_order or b is None) else b.detach().requires_grad_(need_b)
            yr = _chain(xr, br, fu, fd, ctx.cfg)
            ins = [t for t, n in ((xr, need_x), (br, need_b)) if n]
            gs = list(torch.autograd.grad(yr, ins, dy, create_graph=second_order))
        gx = gs.pop(0) if need_x else None
        gb = gs.pop(0) if need_b else None
        return gx, gb, None, None, None
