"""CPU: the StyleGAN-operator oracle (oracle/style_ref.py) against the golden vectors written by the reference's own `impl='ref'` functions
(tests/golden/style_ops.npz, oracle/make_golden_style.py), the regeneration of those vectors from the reference when it is present, and the
argument / error behaviour of the host mirrors that needs no GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import style_ref as SR
from oracle import make_golden_style as MGS
from oracle import ref_import

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "style_ops.npz")


def _t(z, k):
    return torch.from_numpy(z[k]) if k in z.files else None


def test_style_oracle_reproduces_the_reference_vectors():
    z = np.load(GOLD)
    for (tag, shape, dim, act, alpha, gain, clamp, wb) in MGS.BIAS_ACT_CASES:
        p = f"bias_act/{tag}/"
        x = _t(z, p + "x").requires_grad_(True)
        b = _t(z, p + "b")
        b = b.requires_grad_(True) if b is not None else None
        y = SR.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
        assert torch.equal(y.detach(), _t(z, p + "y")), tag
        g = torch.autograd.grad(y, [x] + ([b] if wb else []), _t(z, p + "gy"), create_graph=True)
        assert float((g[0].detach() - _t(z, p + "dx")).abs().max()) <= 1e-6, tag
        if wb:
            assert float((g[1].detach() - _t(z, p + "db")).abs().max()) <= 1e-5, tag
        if g[0].requires_grad:
            g2 = torch.autograd.grad(g[0], x, _t(z, p + "gg"), allow_unused=True)[0]
            g2 = torch.zeros_like(x) if g2 is None else g2
            assert float((g2 - _t(z, p + "ddx")).abs().max()) <= 1e-5, tag
    for (tag, shape, taps, sep, up, down, pad, flip, gain) in MGS.UPFIRDN_CASES:
        p = f"upfirdn2d/{tag}/"
        x = _t(z, p + "x").requires_grad_(True)
        y = SR.upfirdn2d(x, _t(z, p + "f"), up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        assert torch.equal(y.detach(), _t(z, p + "y")), tag
        assert float((torch.autograd.grad(y, x, _t(z, p + "gy"))[0] - _t(z, p + "dx")).abs().max()) <= 1e-6, tag
    for (tag, shape, tu, td, up, down, pad, gain, slope, clamp, flip) in MGS.FLRELU_CASES:
        p = f"filtered_lrelu/{tag}/"
        x, b = _t(z, p + "x").requires_grad_(True), _t(z, p + "b").requires_grad_(True)
        y = SR.filtered_lrelu(x, fu=_t(z, p + "fu"), fd=_t(z, p + "fd"), b=b, up=up, down=down, padding=pad, gain=float(gain), slope=slope, clamp=clamp, flip_filter=flip)
        assert torch.equal(y.detach(), _t(z, p + "y")), tag
        gx, gb = torch.autograd.grad(y, [x, b], _t(z, p + "gy"))
        assert float((gx - _t(z, p + "dx")).abs().max()) <= 1e-6 and float((gb - _t(z, p + "db")).abs().max()) <= 1e-5, tag


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_style_golden_vectors_regenerate_from_the_reference(tmp_path, monkeypatch):
    """the committed fixture IS what the reference's ref functions produce today, and the restatement matches them bit for bit"""
    out = tmp_path / "style_ops.npz"
    monkeypatch.setattr(MGS, "OUT", str(out))
    MGS.main()
    a, b = np.load(GOLD), np.load(out)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_style_ops_argument_checks_and_no_cpu_fallback():
    import studiogan_amd  # noqa: F401
    from studiogan_amd.style_ops import bias_act, upfirdn2d, filtered_lrelu
    x = torch.randn(2, 3, 4, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bias_act.bias_act(x, torch.randn(3), act="lrelu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        upfirdn2d.upfirdn2d(x, upfirdn2d.setup_filter([1, 3, 3, 1]), up=2, padding=2)
    with pytest.raises(KeyError):
        bias_act.bias_act(x, act="gelu")
    with pytest.raises(AssertionError):
        bias_act.bias_act(x, clamp=-1.0)
    with pytest.raises(AssertionError):
        upfirdn2d.upfirdn2d(x, None, up=0)
    with pytest.raises(AssertionError):
        filtered_lrelu.filtered_lrelu(x, b=torch.randn(4))
    # the filter helper is pure host code: same constants as the reference's setup_filter on the shapes StyleGAN uses
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    assert f.shape == (4, 4) and abs(float(f.sum()) - 1.0) < 1e-6 and torch.allclose(f, f.t())
    assert upfirdn2d.setup_filter(list(range(1, 13))).dim() == 1          # >= 8 taps stay separable
    assert upfirdn2d.setup_filter([1, 2, 1], gain=4).sum().item() == pytest.approx(4.0)
    # activation table = the reference's (names, defaults)
    assert set(bias_act.activation_funcs) == set(SR.ACTS)
    for k, (fn, da, dg) in SR.ACTS.items():
        assert bias_act.activation_funcs[k].def_alpha == da and bias_act.activation_funcs[k].def_gain == pytest.approx(dg)
