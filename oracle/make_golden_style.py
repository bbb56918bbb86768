"""Golden vectors of the StyleGAN operators written by the REAL reference's own `impl='ref'` functions
(reference src/utils/style_ops/{bias_act,upfirdn2d,filtered_lrelu}.py, imported on CPU through oracle/ref_import.py), and the pin of the
restatement oracle/style_ref.py against them (bit-identical: max error 0). Output: tests/golden/style_ops.npz.

    python -m oracle.make_golden_style           (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import importlib
import os

import numpy as np
import torch

from . import ref_import
from . import style_ref as SR

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "style_ops.npz")

BIAS_ACT_CASES = [   # (tag, shape, dim, act, alpha, gain, clamp, with bias)
    ("lin_b", (3, 5, 4, 6), 1, "linear", None, None, None, True), ("relu", (3, 5, 4, 6), 1, "relu", None, None, None, True),
    ("lrelu_clamp", (2, 7, 5, 3), 1, "lrelu", 0.1, 1.7, 0.9, True), ("lrelu_def", (4, 16), 1, "lrelu", None, None, None, True),
    ("tanh", (2, 6, 3, 3), 1, "tanh", None, 1.3, None, True), ("sigmoid", (2, 6, 3, 3), 1, "sigmoid", None, None, 0.6, False),
    ("elu", (5, 9), 1, "elu", None, None, None, True), ("selu", (5, 9), 0, "selu", None, 0.8, None, True),
    ("softplus", (2, 3, 5, 5), 1, "softplus", None, None, 1.1, True), ("swish", (2, 3, 5, 5), 1, "swish", None, None, 1.2, True),
    ("swish_dim3", (2, 3, 5, 4), 3, "swish", None, 1.0, None, True), ("gain_only", (33,), 0, "linear", None, 2.5, None, False),
]
UPFIRDN_CASES = [    # (tag, shape, filter taps or 2-D, separable?, up, down, padding, flip, gain)
    ("up2", (2, 3, 8, 8), [1, 3, 3, 1], False, 2, 1, [2, 1, 2, 1], False, 4.0),
    ("down2", (2, 3, 9, 11), [1, 3, 3, 1], False, 1, 2, [1, 1, 1, 1], False, 1.0),
    ("sep12_up2", (1, 4, 10, 7), list(np.hanning(14)[1:-1]), True, 2, 1, [6, 5, 6, 5], False, 4.0),
    ("asym_flip", (2, 2, 7, 6), [[1, 2, 0], [3, -1, 4]], False, [2, 1], [1, 3], [1, 2, 0, 3], True, 0.7),
    ("crop", (1, 3, 12, 12), [1, 2, 1], False, 1, 1, [-2, -1, 1, -3], False, 1.0),
    ("updown", (2, 3, 6, 5), [1, 4, 6, 4, 1], False, 3, 2, [4, 4, 3, 5], False, 9.0),
    ("identity", (2, 3, 5, 5), None, False, 1, 1, 0, False, 1.0),
]
FLRELU_CASES = [     # (tag, shape, fu taps, fd taps, up, down, padding, gain, slope, clamp, flip)
    ("sg3", (2, 4, 10, 10), list(np.hanning(14)[1:-1]), list(np.hanning(14)[1:-1]), 2, 2, 11, np.sqrt(2), 0.2, 256.0, False),
    ("up4_down2_clamp", (1, 3, 6, 7), [1, 3, 3, 1], [1, 2, 1], 4, 2, [3, 2, 4, 1], 1.5, 0.1, 0.8, False),
    ("no_filters", (2, 3, 5, 5), None, None, 1, 1, 0, np.sqrt(2), 0.2, None, False),
    ("full2d_flip", (1, 2, 8, 8), [[1, 2, 1], [2, 4, 2], [1, 2, 3]], [[1, 1], [1, 2]], 2, 1, [2, 2, 2, 2], 1.0, 0.3, None, True),
]


def _rand(shape, seed, scale=1.5):
    return (scale * torch.randn(shape, generator=torch.Generator().manual_seed(seed))).float()


def main():
    assert ref_import.available(), "needs the reference checkout"
    ref_import._prepare()
    RB = importlib.import_module("utils.style_ops.bias_act")
    RU = importlib.import_module("utils.style_ops.upfirdn2d")
    RF = importlib.import_module("utils.style_ops.filtered_lrelu")
    out, worst = {}, 0.0

    def cmp(a, b):
        nonlocal worst
        worst = max(worst, float((a.double() - b.double()).abs().max()))

    for i, (tag, shape, dim, act, alpha, gain, clamp, wb) in enumerate(BIAS_ACT_CASES):
        x = _rand(shape, 100 + i).requires_grad_(True)
        b = _rand((shape[dim],), 200 + i, 0.5).requires_grad_(True) if wb else None
        gy, gg = _rand(shape, 300 + i, 1.0), _rand(shape, 400 + i, 1.0)
        res = {}
        for name, fn in (("ref", lambda: RB.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp, impl="ref")),
                         ("mine", lambda: SR.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp))):
            y = fn()
            ins = [x] + ([b] if wb else [])
            g1 = torch.autograd.grad(y, ins, gy, create_graph=True)
            # second order: gradient of <dx, gg> w.r.t. (x, gy-as-constant): d/dx of the first-order gradient
            g2 = torch.autograd.grad(g1[0], x, gg, allow_unused=True)[0] if g1[0].requires_grad else None
            res[name] = (y.detach(), g1[0].detach(), g1[1].detach() if wb else None, torch.zeros_like(x) if g2 is None else g2.detach())
        for a, c in zip(res["ref"], res["mine"]):
            if a is not None:
                cmp(a, c)
        p = f"bias_act/{tag}/"
        out[p + "x"], out[p + "gy"], out[p + "gg"] = x.detach().numpy(), gy.numpy(), gg.numpy()
        if wb:
            out[p + "b"], out[p + "db"] = b.detach().numpy(), res["ref"][2].numpy()
        out[p + "y"], out[p + "dx"], out[p + "ddx"] = res["ref"][0].numpy(), res["ref"][1].numpy(), res["ref"][3].numpy()

    def filt(taps, sep):
        if taps is None:
            return None
        t = torch.tensor(taps, dtype=torch.float32)
        return t if (t.dim() == 2 or sep) else torch.outer(t, t)

    for i, (tag, shape, taps, sep, up, down, pad, flip, gain) in enumerate(UPFIRDN_CASES):
        x = _rand(shape, 500 + i).requires_grad_(True)
        f = filt(taps, sep)
        ya = RU.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl="ref")
        yb = SR.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        gy = _rand(tuple(ya.shape), 600 + i, 1.0)
        da, db = torch.autograd.grad(ya, x, gy)[0], torch.autograd.grad(yb, x, gy)[0]
        cmp(ya.detach(), yb.detach()); cmp(da, db)
        p = f"upfirdn2d/{tag}/"
        out[p + "x"], out[p + "gy"], out[p + "y"], out[p + "dx"] = x.detach().numpy(), gy.numpy(), ya.detach().numpy(), da.numpy()
        if f is not None:
            out[p + "f"] = f.numpy()

    for i, (tag, shape, tu, td, up, down, pad, gain, slope, clamp, flip) in enumerate(FLRELU_CASES):
        x = _rand(shape, 700 + i).requires_grad_(True)
        b = _rand((shape[1],), 800 + i, 0.5).requires_grad_(True)
        fu = None if tu is None else torch.tensor(tu, dtype=torch.float32)
        fd = None if td is None else torch.tensor(td, dtype=torch.float32)
        fu = fu if fu is None else fu / fu.sum()
        fd = fd if fd is None else fd / fd.sum()
        ya = RF.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=pad, gain=gain, slope=slope, clamp=clamp, flip_filter=flip, impl="ref")
        yb = SR.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=pad, gain=float(gain), slope=slope, clamp=clamp, flip_filter=flip)
        gy = _rand(tuple(ya.shape), 900 + i, 1.0)
        ga, gb = torch.autograd.grad(ya, [x, b], gy), torch.autograd.grad(yb, [x, b], gy)
        cmp(ya.detach(), yb.detach()); cmp(ga[0], gb[0]); cmp(ga[1], gb[1])
        p = f"filtered_lrelu/{tag}/"
        out[p + "x"], out[p + "b"], out[p + "gy"] = x.detach().numpy(), b.detach().numpy(), gy.numpy()
        out[p + "y"], out[p + "dx"], out[p + "db"] = ya.detach().numpy(), ga[0].numpy(), ga[1].numpy()
        if fu is not None:
            out[p + "fu"] = fu.numpy()
        if fd is not None:
            out[p + "fd"] = fd.numpy()
    print(f"restatement vs the reference's own ref functions: max |diff| = {worst:.3e} over {len(BIAS_ACT_CASES) + len(UPFIRDN_CASES) + len(FLRELU_CASES)} cases")
    assert worst <= 1e-6, "oracle/style_ref.py disagrees with the reference"
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
