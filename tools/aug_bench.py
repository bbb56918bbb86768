"""sg_augment_fwd / sg_augment_bwd / sg_mse alone at the benchmark's image batch (256 x 3 x 128 x 128 fp32): microseconds per call (hipEvent pairs on the launch
stream) and GB/s of ALGORITHMIC bytes against the 8 TB/s HBM peak. Algorithmic bytes per element: forward 4 (read x) + 4 (write y) + 4 more when the contrast mean
needs its own pass over x; backward the same with dy / dx. The reference's chain for the same policy runs 14 launches (diffaug.py:47-95)."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiogan_amd  # noqa: E402,F401
from studiogan_amd import functional as F, _lib as L, diffaug, cr, losses  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    # (the first shape is measured twice: the first timed loop of a fresh process shows ~0.8 ms per call of host-side one-off cost -- the kernel trace of the same
    #  run has no k_aug_* launch above 24 us, profiles/r05_augment_kerneltrace_k.txt)
    for shape in ((256, 3, 128, 128), (256, 3, 128, 128), (256, 3, 32, 32), (64, 3, 256, 256)):
        N, C, H, W = shape
        n = N * C * H * W
        x = torch.rand(shape, device=dev) * 2 - 1
        g = torch.Generator().manual_seed(0)
        color = torch.stack([torch.rand(N, generator=g) - 0.5, torch.rand(N, generator=g) * 2, torch.rand(N, generator=g) + 0.5], 1).to(dev)
        mt = H // 8
        geom = torch.stack([torch.randint(-mt, mt + 1, (N,), generator=g), torch.randint(-mt, mt + 1, (N,), generator=g), torch.randint(0, H, (N,), generator=g),
                            torch.randint(0, W, (N,), generator=g), torch.randint(0, 2, (N,), generator=g)], 1).to(torch.int32).to(dev)
        full = L.AUG_BRIGHTNESS | L.AUG_SATURATION | L.AUG_CONTRAST | L.AUG_TRANSLATE | L.AUG_CUTOUT
        rows = [("color,translation,cutout (DiffAugment)", full, 12), ("translation,cutout", L.AUG_TRANSLATE | L.AUG_CUTOUT, 8),
                ("flip + reflect translation (CR)", L.AUG_FLIP | L.AUG_TRANSLATE_REFLECT, 8)]
        print(f"--- {shape}: {n * 4 / 1e6:.1f} MB per image batch")
        for name, ops, bpe in rows:
            spec = F.AugSpec(ops, color, geom, H // 2, W // 2, mt)
            with torch.no_grad():
                tf = timed(lambda: F.AugmentFn.apply(x, spec))
                tb = timed(lambda: F.AugmentBwdFn.apply(x, spec))
            print(f"{name:42s} fwd {tf:8.1f} us {n * bpe / tf / 1e3:7.1f} GB/s ({n * bpe / tf / 1e3 / 8000:.3f} of 8 TB/s)   bwd {tb:8.1f} us {n * bpe / tb / 1e3:7.1f} GB/s "
                  f"({n * bpe / tb / 1e3 / 8000:.3f})   [incl. the torch.empty of the result]")
        with torch.no_grad():
            th = timed(lambda: diffaug.apply_diffaug(x))
            tc = timed(lambda: cr.apply_cr_aug(x))
            y = torch.rand_like(x)
            tm = timed(lambda: losses.l2_loss(x, y))
        print(f"{'apply_diffaug (draws + tables + launch)':42s} {th:8.1f} us    apply_cr_aug {tc:8.1f} us    l2_loss {tm:8.1f} us = {n * 8 / tm / 1e3:7.1f} GB/s")
        from studiogan_amd import ada_aug
        aug = ada_aug.AdaAugment(**ada_aug.AUGPIPE["bgc"]).to(dev)
        aug.p.copy_(torch.as_tensor(0.6))
        with torch.no_grad():
            ta = timed(lambda: aug(x), reps=20)
        xg = x.clone().requires_grad_(True)
        tab = timed(lambda: torch.autograd.grad(aug(xg).sum(), xg), reps=20)
        print(f"{'AdaAugment bgc, p = 0.6 (whole module)':42s} fwd {ta:8.1f} us    fwd + bwd {tab:8.1f} us")


if __name__ == "__main__":
    main()
