"""Golden vectors of the image augmentations written by the REAL reference's own functions (reference src/utils/diffaug.py apply_diffaug,
src/utils/cr.py apply_cr_aug, torch.nn.MSELoss as src/worker.py:116 builds it), imported on CPU through oracle/ref_import.py under a seeded
generator, and the pin of the restatement oracle/aug_ref.py (draws as arguments) against them. Output: tests/golden/aug.npz.

    python -m oracle.make_golden_aug           (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import importlib
import os

import numpy as np
import torch

from . import aug_ref as AR
from . import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aug.npz")

DIFFAUG_CASES = [   # (tag, shape, policy)
    ("full32", (4, 3, 32, 32), "color,translation,cutout"), ("full_rect", (3, 3, 16, 24), "color,translation,cutout"),
    ("odd", (2, 3, 7, 9), "color,translation,cutout"), ("gray", (3, 1, 8, 8), "color,translation,cutout"),
    ("geo", (4, 3, 16, 16), "translation,cutout"), ("color", (4, 3, 12, 12), "color"), ("cut_first", (3, 3, 16, 16), "cutout,color"),
    ("trans", (5, 3, 8, 8), "translation"), ("twice", (2, 3, 16, 16), "translation,translation,cutout,cutout"), ("two_ch", (2, 2, 6, 10), "color,cutout"),
]
CR_CASES = [        # (tag, shape, flip, translation)
    ("both32", (6, 3, 32, 32), True, True), ("flip", (5, 3, 8, 12), True, False), ("trans_rect", (4, 3, 16, 24), False, True), ("odd", (3, 3, 9, 17), True, True),
]
MSE_CASES = [("logits", (64,)), ("embed", (16, 48)), ("images", (4, 3, 32, 32))]
LOSS_KINDS = ["least_square", "logistic", "vanilla"]           # (vanilla rides along: the logistic mirror is the vanilla kernel)
FM_CASES = [("narrow", (4, 16)), ("wide", (64, 1536)), ("ragged", (7, 300))]


def _rand(shape, seed, scale=0.8):
    return (scale * torch.randn(shape, generator=torch.Generator().manual_seed(seed))).float()


def main():
    assert ref_import.available(), "needs the reference checkout"
    ref_import._prepare()
    RD = importlib.import_module("utils.diffaug")
    RC = importlib.import_module("utils.cr")
    out, worst = {}, 0.0
    for i, (tag, shape, policy) in enumerate(DIFFAUG_CASES):
        x = _rand(shape, 1000 + i).clamp(-1, 1).requires_grad_(True)
        gy, gg = _rand(shape, 2000 + i, 1.0), _rand(shape, 3000 + i, 1.0)
        torch.manual_seed(4000 + i)
        draws = AR.draw_diffaug(shape, policy)
        torch.manual_seed(4000 + i)
        y = RD.apply_diffaug(x, policy)                       # the reference, drawing the same numbers from the same generator state
        dx = torch.autograd.grad(y, x, gy)[0]
        mine = AR.diffaug(x, policy, draws)
        assert torch.equal(mine, y), (tag, float((mine - y).abs().max()))
        worst = max(worst, float((torch.autograd.grad(mine, x, gy)[0] - dx).abs().max()))     # (autograd sums the mean's cotangents in another order)
        # second order (what R1 through an augmented batch needs): u -> d/dgy <A^T gy, u> = A u, the linear part of the map
        gyv = gy.clone().requires_grad_(True)
        dxv = torch.autograd.grad(RD_apply_seeded(RD, x, policy, 4000 + i), x, gyv, create_graph=True)[0]
        lin = torch.autograd.grad(dxv, gyv, gg)[0]
        p = f"diffaug/{tag}/"
        out[p + "x"], out[p + "gy"], out[p + "gg"] = x.detach().numpy(), gy.numpy(), gg.numpy()
        for k, d in enumerate(draws):
            out[p + f"draw{k}"] = d.numpy()
        out[p + "y"], out[p + "dx"], out[p + "lin"] = y.detach().numpy(), dx.numpy(), lin.numpy()
    for i, (tag, shape, flip, trans) in enumerate(CR_CASES):
        x = _rand(shape, 5000 + i).clamp(-1, 1).requires_grad_(True)
        gy = _rand(shape, 6000 + i, 1.0)
        torch.manual_seed(7000 + i)
        coin, tx, ty = AR.draw_cr(shape, flip, trans)
        torch.manual_seed(7000 + i)
        y = RC.apply_cr_aug(x, flip=flip, translation=trans)
        dx = torch.autograd.grad(y, x, gy)[0]
        mine = AR.cr_aug(x, coin, tx, ty)
        assert torch.equal(mine, y), tag
        worst = max(worst, float((torch.autograd.grad(mine, x, gy)[0] - dx).abs().max()))
        p = f"cr/{tag}/"
        out[p + "x"], out[p + "gy"], out[p + "y"], out[p + "dx"] = x.detach().numpy(), gy.numpy(), y.detach().numpy(), dx.numpy()
        if flip:
            out[p + "coin"] = coin.numpy()
        if trans:
            out[p + "tx"], out[p + "ty"] = tx.numpy(), ty.numpy()
    l2 = torch.nn.MSELoss()                                   # reference src/worker.py:116
    for i, (tag, shape) in enumerate(MSE_CASES):
        a, b = _rand(shape, 8000 + i).requires_grad_(True), _rand(shape, 8100 + i).requires_grad_(True)
        loss = l2(a, b)
        da, db = torch.autograd.grad(loss, [a, b], torch.tensor(0.7))
        worst = max(worst, float((AR.mse(a, b) - loss).detach().abs()))
        p = f"mse/{tag}/"
        out[p + "a"], out[p + "b"], out[p + "loss"], out[p + "da"], out[p + "db"] = a.detach().numpy(), b.detach().numpy(), loss.detach().numpy(), da.numpy(), db.numpy()
    RL = importlib.import_module("utils.losses")
    from . import restate as O
    for i, kind in enumerate(LOSS_KINDS):                     # src/utils/losses.py:197-223 through the names src/config.py:411-433 binds
        r, f = _rand((37,), 9000 + i, 1.5).requires_grad_(True), _rand((37,), 9100 + i, 1.5).requires_grad_(True)
        short = {"least_square": "ls"}.get(kind, kind)
        dl = getattr(RL, "d_" + short)(r, f, DDP=False)
        dr, df = torch.autograd.grad(dl, [r, f], torch.tensor(1.3))
        gl = getattr(RL, "g_" + short)(f, DDP=False)
        (gf,) = torch.autograd.grad(gl, f, torch.tensor(1.3))
        worst = max(worst, float((O.d_loss(kind, r, f) - dl).detach().abs()), float((O.g_loss(kind, f) - gl).detach().abs()))
        p = f"loss/{kind}/"
        out[p + "real"], out[p + "fake"] = r.detach().numpy(), f.detach().numpy()
        out[p + "d"], out[p + "d_dreal"], out[p + "d_dfake"], out[p + "g"], out[p + "g_dfake"] = dl.detach().numpy(), dr.numpy(), df.numpy(), gl.detach().numpy(), gf.numpy()
    for i, (tag, shape) in enumerate(FM_CASES):               # src/utils/losses.py:254-259
        hr, hf = _rand(shape, 9200 + i), _rand(shape, 9300 + i).requires_grad_(True)
        fm = RL.feature_matching_loss(hr, hf)
        (dh,) = torch.autograd.grad(fm, hf, torch.tensor(0.9))
        worst = max(worst, float((O.feature_matching(hr, hf) - fm).detach().abs()))
        p = f"fm/{tag}/"
        out[p + "real"], out[p + "fake"], out[p + "loss"], out[p + "dfake"] = hr.numpy(), hf.detach().numpy(), fm.detach().numpy(), dh.numpy()
    RA = importlib.import_module("utils.apa_aug")              # src/utils/apa_aug.py:10-21
    for i, pr in enumerate((0.0, 0.5, 1.0)):
        real, fake = _rand((6, 3, 8, 8), 9400 + i), _rand((6, 3, 8, 8), 9500 + i)
        torch.manual_seed(9600 + i)
        coin = torch.rand([6, 1, 1, 1])
        torch.manual_seed(9600 + i)
        y = RA.apply_apa_aug(real, fake, pr, "cpu")
        assert torch.equal(AR.apa(real, fake, coin, pr), y), pr
        p = f"apa/{i}/"
        out[p + "real"], out[p + "fake"], out[p + "coin"], out[p + "p"], out[p + "y"] = real.numpy(), fake.numpy(), coin.numpy(), np.float32(pr), y.numpy()
    assert worst <= 1e-6, worst
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "restatement bit-identical to the reference on", len(DIFFAUG_CASES) + len(CR_CASES), "augmentation cases (outputs); gradients / mse worst", worst)


def RD_apply_seeded(RD, x, policy, seed):
    torch.manual_seed(seed)
    return RD.apply_diffaug(x, policy)


if __name__ == "__main__":
    main()
