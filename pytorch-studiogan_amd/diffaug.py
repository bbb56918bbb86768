"""DiffAugment with the reference's name, signature and random-number consumption (reference src/utils/diffaug.py:35-102; wired as
cfgs.AUG.series_augment by src/config.py:586-587 and as the CR / bCR `parallel_augment` by :605-606,619-620; called on the real and the fake batch
in front of every discriminator forward, src/worker.py:276-278,549-550).

The reference runs each operator as its own chain of torch kernels (and materialises an int64 index grid per translation / cutout). Here the
operators of a policy are gathered into sg_augment calls (csrc/ext/augment.hip: ONE gather pass per call, one extra partial-sum pass for the
contrast mean); the per-image random draws are made exactly as the reference makes them -- same torch calls, same order, same device -- so a
seeded run consumes the generator's stream identically. Differentiable to second order (functional.AugmentFn / AugmentBwdFn).
"""
import torch

from . import _lib as L
from . import functional as F

# operator -> (kernel bit, position in the kernel's fixed order)
_ORDER = {"brightness": (L.AUG_BRIGHTNESS, 0), "saturation": (L.AUG_SATURATION, 1), "contrast": (L.AUG_CONTRAST, 2),
          "translation": (L.AUG_TRANSLATE, 4), "cutout": (L.AUG_CUTOUT, 5)}
AUGMENT_FNS = {"color": ["brightness", "saturation", "contrast"], "translation": ["translation"], "cutout": ["cutout"]}   # diffaug.py:98-102


class _Group:
    """operators collected for one kernel call, with their draws"""

    def __init__(self, x):
        self.x, self.ops, self.last = x, 0, -1
        self.b = self.s = self.c = self.tx = self.ty = self.cx = self.cy = None
        self.cut = (0, 0)

    def spec(self):
        N, dev = self.x.shape[0], self.x.device
        color = geom = None
        if self.ops & (L.AUG_BRIGHTNESS | L.AUG_SATURATION | L.AUG_CONTRAST):
            cols = [t.reshape(N).float() if t is not None else torch.full((N,), fill, dtype=torch.float32, device=dev)
                    for t, fill in ((self.b, 0.0), (self.s, 1.0), (self.c, 1.0))]
            color = torch.stack(cols, 1).contiguous()
        if self.ops & (L.AUG_TRANSLATE | L.AUG_CUTOUT):
            z = torch.zeros(N, dtype=torch.long, device=dev)
            cols = [t.reshape(N) if t is not None else z for t in (self.tx, self.ty, self.cx, self.cy)] + [z]
            geom = torch.stack(cols, 1).to(torch.int32).contiguous()
        return F.AugSpec(self.ops, color, geom, self.cut[0], self.cut[1])


def _flush(g):
    return F.AugmentFn.apply(g.x, g.spec()) if g.ops else g.x


def apply_diffaug(x, policy="color,translation,cutout", channels_first=True):
    """reference src/utils/diffaug.py:35-45"""
    if policy:
        L.require_gpu(x.device)                 # no CPU fallback on the product path
        if not channels_first:
            x = x.permute(0, 3, 1, 2)
        N, _, H, W = x.shape
        g = _Group(x)
        for p in policy.split(","):
            for name in AUGMENT_FNS[p]:
                bit, pos = _ORDER[name]
                if pos <= g.last:               # out of the kernel's fixed order (or a repeat): close the call, open the next on its result
                    g = _Group(_flush(g))
                g.ops, g.last = g.ops | bit, pos
                # the draws below are the reference's own calls (diffaug.py:48,54,60,66-67,81-82), in its order
                if name == "brightness":
                    g.b = torch.rand(N, 1, 1, 1, dtype=x.dtype, device=x.device) - 0.5
                elif name == "saturation":
                    g.s = torch.rand(N, 1, 1, 1, dtype=x.dtype, device=x.device) * 2
                elif name == "contrast":
                    g.c = torch.rand(N, 1, 1, 1, dtype=x.dtype, device=x.device) + 0.5
                elif name == "translation":
                    shift_x, shift_y = int(H * 0.125 + 0.5), int(W * 0.125 + 0.5)
                    g.tx = torch.randint(-shift_x, shift_x + 1, size=[N, 1, 1], device=x.device)
                    g.ty = torch.randint(-shift_y, shift_y + 1, size=[N, 1, 1], device=x.device)
                else:
                    g.cut = (int(H * 0.5 + 0.5), int(W * 0.5 + 0.5))
                    g.cx = torch.randint(0, H + (1 - g.cut[0] % 2), size=[N, 1, 1], device=x.device)
                    g.cy = torch.randint(0, W + (1 - g.cut[1] % 2), size=[N, 1, 1], device=x.device)
        x = _flush(g)
        if not channels_first:
            x = x.permute(0, 2, 3, 1)
        x = x.contiguous()
    return x
