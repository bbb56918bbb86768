"""CPU restatement of the reference's StyleGAN operators (reference src/utils/style_ops/bias_act.py:89-120 `_bias_act_ref`,
upfirdn2d.py:166-213 `_upfirdn2d_ref`, filtered_lrelu.py:120-155 `_filtered_lrelu_ref`) in plain torch CPU ops.

TEST INFRASTRUCTURE: the checker of tests/test_style_gpu.py (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/). Pinned against the reference's own functions by oracle/make_golden_style.py (max error 0 at generation time) and, through
tests/golden/style_ops.npz, on every CPU test run. Differentiable through torch autograd (first and second order), which is what the gradient
checks of the HIP operators compare with.
"""
import math

import torch
import torch.nn.functional as TF

SQRT2 = math.sqrt(2.0)
# name -> (function, default alpha, default gain)       (bias_act.py:20-30)
ACTS = {
    "linear": (lambda x, a: x, 0.0, 1.0),
    "relu": (lambda x, a: torch.relu(x), 0.0, SQRT2),
    "lrelu": (lambda x, a: TF.leaky_relu(x, a), 0.2, SQRT2),
    "tanh": (lambda x, a: torch.tanh(x), 0.0, 1.0),
    "sigmoid": (lambda x, a: torch.sigmoid(x), 0.0, 1.0),
    "elu": (lambda x, a: TF.elu(x), 0.0, 1.0),
    "selu": (lambda x, a: TF.selu(x), 0.0, 1.0),
    "softplus": (lambda x, a: TF.softplus(x), 0.0, 1.0),
    "swish": (lambda x, a: torch.sigmoid(x) * x, 0.0, SQRT2),
}


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    """bias_act.py:89-120: bias along `dim`, activation, gain (skipped when 1), symmetric clamp (skipped when negative / None)."""
    fn, da, dg = ACTS[act]
    alpha = float(da if alpha is None else alpha)
    gain = float(dg if gain is None else gain)
    if b is not None:
        shape = [1] * x.dim()
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _pad4(p):
    if isinstance(p, int):
        return p, p, p, p
    p = [int(v) for v in p]
    return (p[0], p[0], p[1], p[1]) if len(p) == 2 else tuple(p)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:166-213. x [N,C,H,W]; f None / [taps] (separable) / [fh,fw]. Zero insertion, pad (negative = crop), correlation with the
    flipped filter (i.e. convolution) unless flip_filter, gain ** (f.ndim / 2) per pass, decimation."""
    N, C, H, W = x.shape
    upx, upy = _pair(up)
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    # zero insertion: sample (h, w) lands on (h * upy, w * upx) of an [H * upy, W * upx] grid
    xu = x.new_zeros((N, C, H * upy, W * upx))
    xu[:, :, ::upy, ::upx] = x
    xu = TF.pad(xu, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    xu = xu[:, :, max(-py0, 0): xu.shape[2] - max(-py1, 0), max(-px0, 0): xu.shape[3] - max(-px1, 0)]
    f = (f * (gain ** (f.dim() / 2))).to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.dim())))
    if f.dim() == 2:
        y = TF.conv2d(xu, f[None, None].repeat(C, 1, 1, 1), groups=C)
    else:
        y = TF.conv2d(xu, f[None, None, None, :].repeat(C, 1, 1, 1), groups=C)
        y = TF.conv2d(y, f[None, None, :, None].repeat(C, 1, 1, 1), groups=C)
    return y[:, :, ::downy, ::downx]


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=SQRT2, slope=0.2, clamp=None, flip_filter=False):
    """filtered_lrelu.py:120-155: bias -> upfirdn2d(fu, up, padding, gain = up^2) -> lrelu(slope) * gain, clamp -> upfirdn2d(fd, down)."""
    px0, px1, py0, py1 = _pad4(padding)
    y = bias_act(x, b)
    y = upfirdn2d(y, fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    y = bias_act(y, act="lrelu", alpha=slope, gain=gain, clamp=clamp)
    return upfirdn2d(y, fd, down=down, flip_filter=flip_filter)
