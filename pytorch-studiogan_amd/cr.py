"""The consistency-regularisation augmentation with the reference's name and signature (reference src/utils/cr.py:13-48; cfgs.AUG.parallel_augment
of src/config.py:600-604,614-618; called at src/worker.py:326-354): per-image horizontal flip with probability 1/2, then an integer translation of
up to 1/8 of the image over a reflect-padded copy. One sg_augment launch (csrc/ext/augment.hip) instead of clone + masked flip + pad + index grid +
gather; the draws are the reference's own calls in its order (the flip coin on the CPU generator, the shifts on the device)."""
import torch

from . import _lib as L
from . import functional as F


def apply_cr_aug(x, flip=True, translation=True):
    """reference src/utils/cr.py:13-20"""
    if not (flip or translation):
        return x
    L.require_gpu(x.device)                     # no CPU fallback on the product path
    N, _, H, W = x.shape
    ops, max_t = 0, 0
    z = torch.zeros(N, dtype=torch.long, device=x.device)
    fl, tx, ty = z, z, z
    if flip:                                    # cr.py:24-31 (p = 0.5)
        ops |= L.AUG_FLIP
        fl = (torch.FloatTensor(N, 1).uniform_(0.0, 1.0).to(x.device) < 0.5).reshape(N).long()
    if translation:                             # cr.py:33-48 (ratio 1/8)
        ops |= L.AUG_TRANSLATE_REFLECT
        max_t_x, max_t_y = int(H * (1 / 8)), int(W * (1 / 8))
        tx = torch.randint(-max_t_x, max_t_x + 1, size=[N, 1, 1], device=x.device).reshape(N)
        ty = torch.randint(-max_t_y, max_t_y + 1, size=[N, 1, 1], device=x.device).reshape(N)
        max_t = max(max_t_x, max_t_y)
    geom = torch.stack([tx, ty, z, z, fl], 1).to(torch.int32).contiguous()
    return F.AugmentFn.apply(x, F.AugSpec(ops, None, geom, 0, 0, max_t)).contiguous()
