"""upfirdn2d: pad -> upsample (zero insertion) -> FIR filter -> downsample, and the resampling helpers built on it
(reference src/utils/style_ops/upfirdn2d.py:70-388; plugin contract upfirdn2d.cpp:25-100 -> csrc/style.hip sg_upfirdn2d).
Tensors are NCHW like the reference's; every (n, c) plane is filtered independently."""
import numpy as np
import torch

from .. import _lib as L


def _ints(value, what, allowed):
    """an int or a sequence of ints -> tuple, its length one of `allowed` (a lone int counts as length 1)"""
    seq = (value,) if isinstance(value, int) else value
    ok = isinstance(seq, (list, tuple)) and len(seq) in allowed and all(isinstance(v, int) and not isinstance(v, bool) for v in seq)
    if not ok:
        raise AssertionError(f"{what}: an int or {' / '.join(str(a) for a in allowed if a > 1)} ints expected, got {value!r}")
    return tuple(seq)


def _parse_scaling(scaling):
    """-> (x factor, y factor), each >= 1 (the plugin's own check: upfirdn2d.cpp:35-36)"""
    v = _ints(scaling, "up / down factor", (1, 2))
    sx, sy = v * 2 if len(v) == 1 else v
    if min(sx, sy) < 1:
        raise AssertionError("up / down factors must be >= 1")
    return sx, sy


def _parse_padding(padding):
    """-> (padx0, padx1, pady0, pady1) from one value, (x, y) or all four"""
    v = _ints(padding, "padding", (1, 2, 4))
    return {1: lambda a: (a[0],) * 4, 2: lambda a: (a[0], a[0], a[1], a[1]), 4: tuple}[len(v)](v)


def _get_filter_size(f):
    """(taps along x, taps along y) of a filter constant; None is the 1 x 1 identity"""
    if f is None:
        return 1, 1
    if not isinstance(f, torch.Tensor) or f.dim() not in (1, 2) or f.numel() == 0:
        raise AssertionError("filter must be a non-empty 1-D or 2-D tensor")
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device("cpu"), normalize=True, flip_filter=False, gain=1, separable=None):
    """Filter constant for upfirdn2d with the reference's contract (upfirdn2d.py:70-114): a float32 tensor that is 1-D when the filter is applied separably
    (the default for vectors of >= 8 taps) and the 2-D outer product otherwise; unit sum when `normalize`; `gain` enters as gain ** (ndim / 2) so that a separable
    filter, applied once per axis, carries it exactly once."""
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if taps.dim() > 2 or taps.numel() == 0:
        raise AssertionError("filter must be a non-empty scalar, vector or matrix")
    taps = taps.reshape(1) if taps.dim() == 0 else taps
    keep_1d = (taps.dim() == 1 and taps.numel() >= 8) if separable is None else bool(separable)
    if keep_1d and taps.dim() != 1:
        raise AssertionError("a separable filter is given by its 1-D taps")
    if not keep_1d and taps.dim() == 1:
        taps = torch.outer(taps, taps)
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = torch.flip(taps, dims=tuple(range(taps.dim())))
    return (taps * float(gain) ** (taps.dim() / 2)).to(device=device)


def _launch(x, f2d, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain):
    """one sg_upfirdn2d launch; x: dense NCHW fp32 / bf16 on the GPU; f2d: [fh][fw] fp32"""
    L.require_gpu(x.device)          # the HIP kernels need a GPU tensor: no CPU fallback on the product path
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"upfirdn2d: float32 / bfloat16 only, got {x.dtype}")
    x = x.contiguous()
    N, C, H, W = x.shape
    fh, fw = int(f2d.shape[0]), int(f2d.shape[1])
    Wo = (W * upx + padx0 + padx1 - fw + downx) // downx
    Ho = (H * upy + pady0 + pady1 - fh + downy) // downy
    if Wo < 1 or Ho < 1:
        raise RuntimeError("upfirdn2d: the upsampled and padded image is smaller than the filter")      # upfirdn2d.cpp:49-51
    y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device)
    f2d = f2d.to(device=x.device, dtype=torch.float32).contiguous()
    if N * C > 0:
        L.call("sg_upfirdn2d", L.dt(x), L.ptr(x), L.ptr(f2d), L.ptr(y), N * C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
               1 if flip_filter else 0, float(gain), L.stream())
    return y


_cache = {}


def _make(up, down, padding, flip_filter, gain):
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    key = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
    if key in _cache:
        return _cache[key]

    class Upfirdn2d(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            if not (isinstance(x, torch.Tensor) and x.dim() == 4):
                raise AssertionError("upfirdn2d: x must be a 4-D NCHW tensor")
            if f is None:
                f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            if f.dim() == 1 and f.shape[0] == 1:
                f = f.square().unsqueeze(0)        # a one-tap separable filter is the 1 x 1 filter f^2
            if not (f.dim() in (1, 2) and f.dtype == torch.float32):
                raise AssertionError("upfirdn2d: f must be a float32 vector or matrix")
            if f.dim() == 2:
                y = _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
            else:                                  # separable: a horizontal and a vertical pass (upfirdn2d.py:233-235)
                y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, padx0, padx1, 0, 0, flip_filter, 1.0)
                y = _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, pady0, pady1, flip_filter, gain)
            ctx.save_for_backward(f)
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy):
            (f,) = ctx.saved_tensors
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            # the adjoint of upfirdn is upfirdn with up <-> down, the filter reversed and this padding (upfirdn2d.py:244-250)
            p = [fw - padx0 - 1, iw * upx - ow * downx + padx0 - upx + 1, fh - pady0 - 1, ih * upy - oh * downy + pady0 - upy + 1]
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _make(up=[downx, downy], down=[upx, upy], padding=p, flip_filter=(not flip_filter), gain=gain).apply(dy, f)
            if ctx.needs_input_grad[1]:
                raise AssertionError("upfirdn2d: the filter is a constant (no gradient)")
            return dx, None

    _cache[key] = Upfirdn2d
    return Upfirdn2d


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Pad, upsample, filter and downsample a batch of 2-D images (reference upfirdn2d.py:118-160). Pixels outside the image are zero;
    negative padding crops; without flip_filter the filter is applied as a true convolution."""
    if not isinstance(x, torch.Tensor):
        raise AssertionError("upfirdn2d: x must be a tensor")
    if impl not in ("ref", "cuda"):
        raise AssertionError("upfirdn2d: impl must be 'ref' or 'cuda'")
    return _make(up, down, padding, flip_filter, gain).apply(x, f)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """FIR-filter a batch of images, output padded to the input size (reference upfirdn2d.py:264-298)."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Upsample by an integer factor through the filter, output size a multiple of the input (reference upfirdn2d.py:302-339)."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Downsample by an integer factor through the filter, output size a fraction of the input (reference upfirdn2d.py:343-388)."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
