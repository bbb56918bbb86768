"""GPU: the StyleGAN operators (csrc/style.hip behind studiogan_amd.style_ops) against
  * the golden vectors the reference's own `impl='ref'` functions wrote (tests/golden/style_ops.npz): outputs, first-order gradients and,
    for bias_act, the second-order term;
  * oracle/style_ref.py (pinned bit-identically to the reference) on larger StyleGAN-sized shapes, channels_last tensors and bf16.
fp32 tolerance 2e-5 of the expected tensor's range (the kernels use the hardware exp / log approximations, like the reference's
--use_fast_math build); bf16: 1e-2 against the fp32 oracle evaluated on the bf16-rounded inputs."""
import os

import numpy as np
import pytest
import torch

from util import check
from oracle import style_ref as SR
from oracle import make_golden_style as MGS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "style_ops.npz")
DEV = "cuda:0"


def _t(z, k):
    return torch.from_numpy(z[k]) if k in z.files else None


@pytest.mark.parametrize("case", MGS.BIAS_ACT_CASES, ids=[c[0] for c in MGS.BIAS_ACT_CASES])
def test_bias_act_matches_reference_vectors(sg, case):
    from studiogan_amd.style_ops import bias_act as BA
    tag, shape, dim, act, alpha, gain, clamp, wb = case
    z = np.load(GOLD)
    p = f"bias_act/{tag}/"
    x = _t(z, p + "x").to(DEV).requires_grad_(True)
    b = _t(z, p + "b").to(DEV).requires_grad_(True) if wb else None
    y = BA.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    check(f"bias_act {tag} y", y, _t(z, p + "y"), 2e-5)
    g = torch.autograd.grad(y, [x] + ([b] if wb else []), _t(z, p + "gy").to(DEV), create_graph=True)
    check(f"bias_act {tag} dx", g[0], _t(z, p + "dx"), 2e-5)
    if wb:
        check(f"bias_act {tag} db", g[1], _t(z, p + "db"), 2e-5)
    exp2 = _t(z, p + "ddx")
    if g[0].requires_grad:
        g2 = torch.autograd.grad(g[0], x, _t(z, p + "gg").to(DEV), allow_unused=True)[0]
        g2 = torch.zeros_like(x) if g2 is None else g2
        if float(exp2.abs().max()) > 0:
            check(f"bias_act {tag} second order", g2, exp2, 5e-5)
        else:
            assert float(g2.abs().max()) == 0.0
    else:
        assert float(exp2.abs().max()) == 0.0


@pytest.mark.parametrize("case", MGS.UPFIRDN_CASES, ids=[c[0] for c in MGS.UPFIRDN_CASES])
def test_upfirdn2d_matches_reference_vectors(sg, case):
    from studiogan_amd.style_ops import upfirdn2d as UF
    tag, shape, taps, sep, up, down, pad, flip, gain = case
    z = np.load(GOLD)
    p = f"upfirdn2d/{tag}/"
    x = _t(z, p + "x").to(DEV).requires_grad_(True)
    f = _t(z, p + "f")
    f = f.to(DEV) if f is not None else None
    y = UF.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    exp = _t(z, p + "y")
    assert tuple(y.shape) == tuple(exp.shape)
    check(f"upfirdn2d {tag} y", y, exp, 2e-5)
    gy = _t(z, p + "gy").to(DEV).requires_grad_(True)
    dx = torch.autograd.grad(y, x, gy, create_graph=True)[0]
    check(f"upfirdn2d {tag} dx", dx, _t(z, p + "dx"), 2e-5)
    # the operator is linear: the gradient of <dx, r> w.r.t. gy is the forward operator applied to r
    r = torch.randn(x.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    ggy = torch.autograd.grad(dx, gy, r)[0]
    ref = SR.upfirdn2d(r.cpu(), None if f is None else f.cpu(), up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    check(f"upfirdn2d {tag} double backward", ggy, ref, 2e-5)


@pytest.mark.parametrize("case", MGS.FLRELU_CASES, ids=[c[0] for c in MGS.FLRELU_CASES])
def test_filtered_lrelu_matches_reference_vectors(sg, case):
    from studiogan_amd.style_ops import filtered_lrelu as FL
    tag, shape, tu, td, up, down, pad, gain, slope, clamp, flip = case
    z = np.load(GOLD)
    p = f"filtered_lrelu/{tag}/"
    x, b = _t(z, p + "x").to(DEV).requires_grad_(True), _t(z, p + "b").to(DEV).requires_grad_(True)
    fu, fd = _t(z, p + "fu"), _t(z, p + "fd")
    y = FL.filtered_lrelu(x, fu=None if fu is None else fu.to(DEV), fd=None if fd is None else fd.to(DEV), b=b, up=up, down=down, padding=pad,
                          gain=float(gain), slope=slope, clamp=clamp, flip_filter=flip)
    exp = _t(z, p + "y")
    assert tuple(y.shape) == tuple(exp.shape)
    check(f"filtered_lrelu {tag} y", y, exp, 3e-5)
    gx, gb = torch.autograd.grad(y, [x, b], _t(z, p + "gy").to(DEV))
    check(f"filtered_lrelu {tag} dx", gx, _t(z, p + "dx"), 3e-5)
    check(f"filtered_lrelu {tag} db", gb, _t(z, p + "db"), 3e-5)
    # the one-launch forward (separable filters) against the four-launch operator chain
    FL._FUSED[0] = False
    try:
        y2 = FL.filtered_lrelu(x, fu=None if fu is None else fu.to(DEV), fd=None if fd is None else fd.to(DEV), b=b, up=up, down=down, padding=pad,
                               gain=float(gain), slope=slope, clamp=clamp, flip_filter=flip)
    finally:
        FL._FUSED[0] = True
    check(f"filtered_lrelu {tag} fused vs chain", y, y2.detach().cpu(), 3e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_style_ops_at_stylegan_sizes_vs_oracle(sg, dtype):
    """StyleGAN2 / 3 sized tensors: [4, 64, 64, 64] through bias_act (contiguous and channels_last), a 2x FIR upsample with the 4-tap
    binomial filter, a 12-tap separable up / down pair, and the filtered leaky ReLU; bf16 inputs / outputs with fp32 accumulation."""
    from studiogan_amd.style_ops import bias_act as BA, upfirdn2d as UF, filtered_lrelu as FL
    g = torch.Generator().manual_seed(17)
    x = torch.randn(4, 64, 64, 64, generator=g).to(dtype)
    b = (0.3 * torch.randn(64, generator=g)).to(dtype)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    xf, bf = x.float(), b.float()
    for fmt in (torch.contiguous_format, torch.channels_last):
        xd = x.to(DEV).contiguous(memory_format=fmt)
        y = BA.bias_act(xd, b.to(DEV), act="lrelu", clamp=2.0)
        assert y.is_contiguous(memory_format=fmt)
        check(f"bias_act lrelu {dtype} {fmt}", y.float(), SR.bias_act(xf, bf, act="lrelu", clamp=2.0), tol)
    f4 = UF.setup_filter([1, 3, 3, 1])
    y = UF.upsample2d(x.to(DEV), f4.to(DEV), up=2)
    pad = [(4 + 2 - 1) // 2, (4 - 2) // 2] * 2
    check(f"upsample2d {dtype}", y.float(), SR.upfirdn2d(xf, f4, up=2, padding=pad, gain=4), tol)
    f12 = UF.setup_filter(list(np.hanning(14)[1:-1]))
    assert f12.dim() == 1
    y = UF.downsample2d(x.to(DEV), f12.to(DEV), down=2)
    pd = [(12 - 2 + 1) // 2, (12 - 2) // 2] * 2
    check(f"downsample2d separable {dtype}", y.float(), SR.upfirdn2d(xf, f12, down=2, padding=pd), tol)
    y = FL.filtered_lrelu(x.to(DEV), fu=f12.to(DEV), fd=f12.to(DEV), b=b.to(DEV), up=2, down=2, padding=11, clamp=256.0)
    check(f"filtered_lrelu {dtype}", y.float(), SR.filtered_lrelu(xf, fu=f12, fd=f12, b=bf, up=2, down=2, padding=11, clamp=256.0), 4 * tol)
