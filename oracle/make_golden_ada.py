"""Golden vectors of the adaptive discriminator augmentation pipeline written by the REAL reference's AdaAugment (reference src/utils/ada_aug.py, built with the
`ada_augpipe` entries of src/config.py, imported on CPU through oracle/ref_import.py; its 2x re-sampling runs the reference's own upfirdn2d `impl='ref'`
path): per case the input batch, EVERY random draw the module made (recorded at torch.rand / torch.randn, in consumption order), the output, and the gradient
w.r.t. the images for a fixed cotangent. The product's mirror (studiogan_amd.ada_aug.AdaAugment) is fed the recorded draws and must reproduce output and
gradient (tests/aug_checks.py ada_case). There is no separate restatement: the reference itself is the oracle of this path. Output: tests/golden/ada.npz.

    python -m oracle.make_golden_ada           (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import importlib
import os

import numpy as np
import torch

from . import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ada.npz")

PIPES = {      # reference src/config.py ada_augpipe
    "blit": dict(xflip=1, rotate90=1, xint=1), "geom": dict(scale=1, rotate=1, aniso=1, xfrac=1),
    "color": dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    "bgc": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    "filter": dict(imgfilter=1), "noise": dict(noise=1), "cutout": dict(cutout=1),
    "bgcfnc": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1, imgfilter=1, noise=1,
                   cutout=1),
}
CASES = [      # (tag, pipe, shape, p)
    ("bgc32", "bgc", (4, 3, 32, 32), 0.8), ("bgc_rect", "bgc", (3, 3, 16, 24), 1.0), ("blit16", "blit", (4, 3, 16, 16), 1.0), ("geom16", "geom", (4, 3, 16, 16), 0.9),
    ("color8", "color", (5, 3, 8, 8), 0.8), ("gray", "bgc", (3, 1, 16, 16), 1.0), ("off", "bgc", (2, 3, 8, 8), 0.0),
    ("filter32", "filter", (3, 3, 32, 32), 1.0), ("filter_rect", "filter", (2, 1, 24, 40), 0.8), ("noise8", "noise", (4, 3, 8, 8), 0.9),
    ("cutout16", "cutout", (5, 3, 16, 16), 0.9), ("bgcfnc32", "bgcfnc", (3, 3, 32, 32), 0.9),
]


class Recorded:
    """torch.rand / torch.randn with their results recorded in call order"""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self.saved = (torch.rand, torch.randn)
        r0, n0, draws = torch.rand, torch.randn, self.draws

        def rand(*a, **k):
            t = r0(*a, **k)
            draws.append(t.clone())
            return t

        def randn(*a, **k):
            t = n0(*a, **k)
            draws.append(t.clone())
            return t
        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn = self.saved


def _rand(shape, seed, scale=0.8):
    return (scale * torch.randn(shape, generator=torch.Generator().manual_seed(seed))).float()


def main():
    assert ref_import.available(), "needs the reference checkout"
    ref_import._prepare()
    RA = importlib.import_module("utils.ada_aug")
    out = {}
    for i, (tag, pipe, shape, p) in enumerate(CASES):
        aug = RA.AdaAugment(**PIPES[pipe]).train().requires_grad_(False)
        aug.p.copy_(torch.as_tensor(p))
        x = _rand(shape, 100 + i).clamp(-1, 1).requires_grad_(True)
        gy = _rand(shape, 200 + i, 1.0)
        torch.manual_seed(300 + i)
        with Recorded() as rec:
            y = aug(x)
        (dx,) = torch.autograd.grad(y, x, gy, allow_unused=True)
        pre = f"{tag}/"
        out[pre + "x"], out[pre + "gy"], out[pre + "y"] = x.detach().numpy(), gy.numpy(), y.detach().numpy()
        out[pre + "dx"] = (torch.zeros_like(x) if dx is None else dx).numpy()
        for k, d in enumerate(rec.draws):
            out[pre + f"draw{k}"] = d.numpy()
        print(f"{tag:10s} {pipe:6s} {shape} p={p}: {len(rec.draws)} draws, |y - x| max {float((y.detach() - x.detach()).abs().max()):.3f}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
