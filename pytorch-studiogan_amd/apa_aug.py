"""Adaptive pseudo augmentation with the reference's name and signature (reference src/utils/apa_aug.py:10-21; called on the real batch in front of the
discriminator when AUG.apply_apa, src/worker.py:273-274,459-460): each real image is replaced by the fake image of the same slot with probability apa_p.
One sg_select_rows launch (csrc/ext/regularisers.hip) on the reference's own draw (torch.rand([B, 1, 1, 1]) on the images' device); no host round trip
(the reference's `allclose` early exit returns the real batch itself when no image was drawn -- here the copy holds the same values)."""
import torch

from . import _lib as L
from . import functional as F


def apply_apa_aug(real_images, fake_images, apa_p, local_rank):
    """reference src/utils/apa_aug.py:10-21 (local_rank: the device the draw is made on)"""
    L.require_gpu(real_images.device)
    batch_size = real_images.shape[0]
    flag = torch.rand([batch_size, 1, 1, 1], device=local_rank) < apa_p
    assert fake_images is not None
    return F.select_rows(flag.reshape(batch_size), fake_images, real_images)
