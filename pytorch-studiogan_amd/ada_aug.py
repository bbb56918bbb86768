"""Adaptive discriminator augmentation with the reference's class name, constructor and random-number consumption (reference src/utils/ada_aug.py:102-429
AdaAugment; built by src/config.py:590-591 from the `ada_augpipe` table and used as cfgs.AUG.series_augment, its strength `p` driven by the overfitting
heuristic of src/worker.py:478-487). The pipelines the non-StyleGAN configurations use are mirrored: pixel blitting (x-flip, 90-degree rotations, integer
translation), general geometric transformations (isotropic / anisotropic scaling, pre- and post-rotation, fractional translation), colour transformations
(brightness, contrast, luma flip, hue rotation, saturation), image-space filtering (a per-image 4-band amplification filter), additive noise and cutout: every
entry of the reference's `ada_augpipe` table ('blit', 'geom', 'color', 'filter', 'noise', 'cutout', 'bg', 'bgc', 'bgcf', 'bgcfn', 'bgcfnc'); the non-StyleGAN
configurations (configs/*/{BigGAN,SNGAN,ReACGAN}-ADA.yaml) use 'bgc'.

What the reference does with ~25 tiny launches per draw is kept as it is -- the per-image 3 x 3 / 4 x 4 matrices are composed with torch on [N]-sized tensors,
by the reference's own draw calls in the reference's order -- and everything that touches image bytes runs in libsgamd.so: reflect padding, 2x up-sampling
(sg_upfirdn2d), the affine bilinear resampling WITHOUT a sampling grid in HBM (sg_affine_sample), 2x down-sampling, the colour matrix (sg_color_affine),
each with its exact adjoint (functional.ReflectPad2dFn / AffineSampleFn / ColorAffineFn, style_ops.upfirdn2d)."""
import math

import numpy as np
import torch

from . import _lib as L
from . import functional as F
from .style_ops import upfirdn2d

_SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466, 0.787641141030194, 0.3379294217276218,
         -0.07263752278646252, -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]      # ada_aug.py:37 (the geometric low-pass)

_SYM2 = [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025]      # ada_aug.py:33

AUGPIPE = {      # reference src/config.py: ada_augpipe
    "blit": dict(xflip=1, rotate90=1, xint=1), "geom": dict(scale=1, rotate=1, aniso=1, xfrac=1),
    "color": dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1), "filter": dict(imgfilter=1), "noise": dict(noise=1), "cutout": dict(cutout=1),
    "bg": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1),
    "bgc": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    "bgcf": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1, imgfilter=1),
    "bgcfn": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1, imgfilter=1, noise=1),
    "bgcfnc": dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1, imgfilter=1, noise=1,
                   cutout=1),
}


def _mat(rows, like):
    """[..., r, c] matrix from nested python rows whose entries are floats or tensors of `like`'s shape"""
    elems = [e if torch.is_tensor(e) else torch.full_like(like, float(e)) for row in rows for e in row]
    return torch.stack(elems, dim=-1).reshape(like.shape + (len(rows), len(rows[0])))


def _mm(A, Bm):
    """batched product of small ([..., 3, 3] / [..., 4, 4]) matrices as a broadcast multiply + sum: elementwise launches on [N]-sized tensors, no BLAS call
    on the path (the reference's `@` on these lands in a library GEMM)"""
    return (A.unsqueeze(-1) * Bm.unsqueeze(-3)).sum(-2)


def _translate2d(tx, ty, like):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], like)


def _scale2d(sx, sy, like):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], like)


def _rotate2d(theta):
    c, s = torch.cos(theta), torch.sin(theta)
    return _mat([[c, -s, 0], [s, c, 0], [0, 0, 1]], theta)


class AdaAugment(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1, aniso_std=0.2,
                 xfrac_std=0.125, brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5, hue_max=1,
                 saturation_std=1, imgfilter=0, imgfilter_bands=[1, 1, 1, 1], imgfilter_std=1, noise=0, cutout=0, noise_std=0.1, cutout_size=0.5):
        super().__init__()
        self.register_buffer("p", torch.ones([]))           # overall multiplier of the augmentation probability (ada_aug.py:113)
        for k, v in dict(xflip=xflip, rotate90=rotate90, xint=xint, xint_max=xint_max, scale=scale, rotate=rotate, aniso=aniso, xfrac=xfrac,
                         scale_std=scale_std, rotate_max=rotate_max, aniso_std=aniso_std, xfrac_std=xfrac_std, brightness=brightness, contrast=contrast,
                         lumaflip=lumaflip, hue=hue, saturation=saturation, brightness_std=brightness_std, contrast_std=contrast_std, hue_max=hue_max,
                         saturation_std=saturation_std, imgfilter=imgfilter, imgfilter_std=imgfilter_std, noise=noise, cutout=cutout, noise_std=noise_std,
                         cutout_size=cutout_size).items():
            setattr(self, k, float(v))
        self.imgfilter_bands = list(imgfilter_bands)
        self.register_buffer("Hz_geom", upfirdn2d.setup_filter(_SYM6))
        # filter bank of the image-space filtering (ada_aug.py:166-176): four octave bands built from the sym2 wavelet
        import scipy.signal
        Hz_lo = np.asarray(_SYM2)
        Hz_hi = Hz_lo * ((-1) ** np.arange(Hz_lo.size))
        Hz_lo2 = np.convolve(Hz_lo, Hz_lo[::-1]) / 2
        Hz_hi2 = np.convolve(Hz_hi, Hz_hi[::-1]) / 2
        Hz_fbank = np.eye(4, 1)
        for i in range(1, Hz_fbank.shape[0]):
            Hz_fbank = np.dstack([Hz_fbank, np.zeros_like(Hz_fbank)]).reshape(Hz_fbank.shape[0], -1)[:, :-1]
            Hz_fbank = scipy.signal.convolve(Hz_fbank, [Hz_lo2])
            Hz_fbank[i, (Hz_fbank.shape[1] - Hz_hi2.size) // 2: (Hz_fbank.shape[1] + Hz_hi2.size) // 2] += Hz_hi2
        self.register_buffer("Hz_fbank", torch.as_tensor(Hz_fbank, dtype=torch.float32))

    # ---- parameter selection: the reference's draws, in its order (ada_aug.py:187-262,284-325) --------------------------------------------------------
    def _geometry(self, B, width, height, dev):
        G = None
        one = torch.ones([B], device=dev)

        def mul(G, M):
            return M if G is None else _mm(G, M)
        if self.xflip > 0:
            i = torch.floor(torch.rand([B], device=dev) * 2)
            i = torch.where(torch.rand([B], device=dev) < self.xflip * self.p, i, torch.zeros_like(i))
            G = mul(G, _scale2d(1 / (1 - 2 * i), one, one))
        if self.rotate90 > 0:
            i = torch.floor(torch.rand([B], device=dev) * 4)
            i = torch.where(torch.rand([B], device=dev) < self.rotate90 * self.p, i, torch.zeros_like(i))
            G = mul(G, _rotate2d(np.pi / 2 * i))                      # rotate2d_inv(-pi/2 i) = rotate2d(pi/2 i)
        if self.xint > 0:
            t = (torch.rand([B, 2], device=dev) * 2 - 1) * self.xint_max
            t = torch.where(torch.rand([B, 1], device=dev) < self.xint * self.p, t, torch.zeros_like(t))
            G = mul(G, _translate2d(-torch.round(t[:, 0] * width), -torch.round(t[:, 1] * height), one))
        if self.scale > 0:
            s = torch.exp2(torch.randn([B], device=dev) * self.scale_std)
            s = torch.where(torch.rand([B], device=dev) < self.scale * self.p, s, torch.ones_like(s))
            G = mul(G, _scale2d(1 / s, 1 / s, one))
        p_rot = 1 - torch.sqrt((1 - self.rotate * self.p).clamp(0, 1))
        if self.rotate > 0:
            theta = (torch.rand([B], device=dev) * 2 - 1) * np.pi * self.rotate_max
            theta = torch.where(torch.rand([B], device=dev) < p_rot, theta, torch.zeros_like(theta))
            G = mul(G, _rotate2d(theta))                              # rotate2d_inv(-theta)
        if self.aniso > 0:
            s = torch.exp2(torch.randn([B], device=dev) * self.aniso_std)
            s = torch.where(torch.rand([B], device=dev) < self.aniso * self.p, s, torch.ones_like(s))
            G = mul(G, _scale2d(1 / s, 1 / (1 / s), one))
        if self.rotate > 0:
            theta = (torch.rand([B], device=dev) * 2 - 1) * np.pi * self.rotate_max
            theta = torch.where(torch.rand([B], device=dev) < p_rot, theta, torch.zeros_like(theta))
            G = mul(G, _rotate2d(theta))
        if self.xfrac > 0:
            t = torch.randn([B, 2], device=dev) * self.xfrac_std
            t = torch.where(torch.rand([B, 1], device=dev) < self.xfrac * self.p, t, torch.zeros_like(t))
            G = mul(G, _translate2d(-(t[:, 0] * width), -(t[:, 1] * height), one))
        return G

    def _colour(self, B, C_img, dev):
        Cm = None
        one = torch.ones([B], device=dev)
        I4 = torch.eye(4, device=dev)

        def lmul(M, Cm):
            return M if Cm is None else _mm(M.expand(B, 4, 4) if M.dim() == 2 else M, Cm)
        v = torch.as_tensor(np.asarray([1, 1, 1, 0]) / np.sqrt(3), dtype=torch.float32, device=dev)       # luma axis
        if self.brightness > 0:
            b = torch.randn([B], device=dev) * self.brightness_std
            b = torch.where(torch.rand([B], device=dev) < self.brightness * self.p, b, torch.zeros_like(b))
            Cm = lmul(_mat([[1, 0, 0, b], [0, 1, 0, b], [0, 0, 1, b], [0, 0, 0, 1]], one), Cm)
        if self.contrast > 0:
            c = torch.exp2(torch.randn([B], device=dev) * self.contrast_std)
            c = torch.where(torch.rand([B], device=dev) < self.contrast * self.p, c, torch.ones_like(c))
            Cm = lmul(_mat([[c, 0, 0, 0], [0, c, 0, 0], [0, 0, c, 0], [0, 0, 0, 1]], one), Cm)
        if self.lumaflip > 0:
            i = torch.floor(torch.rand([B, 1, 1], device=dev) * 2)
            i = torch.where(torch.rand([B, 1, 1], device=dev) < self.lumaflip * self.p, i, torch.zeros_like(i))
            Cm = lmul(I4 - 2 * torch.outer(v, v) * i, Cm)             # Householder reflection
        if self.hue > 0 and C_img > 1:
            theta = (torch.rand([B], device=dev) * 2 - 1) * np.pi * self.hue_max
            theta = torch.where(torch.rand([B], device=dev) < self.hue * self.p, theta, torch.zeros_like(theta))
            vx, vy, vz = v[0], v[1], v[2]
            s, c = torch.sin(theta), torch.cos(theta)
            cc = 1 - c
            Cm = lmul(_mat([[vx * vx * cc + c, vx * vy * cc - vz * s, vx * vz * cc + vy * s, 0], [vy * vx * cc + vz * s, vy * vy * cc + c, vy * vz * cc - vx * s, 0],
                            [vz * vx * cc - vy * s, vz * vy * cc + vx * s, vz * vz * cc + c, 0], [0, 0, 0, 1]], one), Cm)
        if self.saturation > 0 and C_img > 1:
            s = torch.exp2(torch.randn([B, 1, 1], device=dev) * self.saturation_std)
            s = torch.where(torch.rand([B, 1, 1], device=dev) < self.saturation * self.p, s, torch.ones_like(s))
            Cm = lmul(torch.outer(v, v) + (I4 - torch.outer(v, v)) * s, Cm)
        return Cm

    # ---- execution ------------------------------------------------------------------------------------------------------------------------------------------
    def _warp(self, images, G_inv):
        """ada_aug.py:250-281: reflect-pad by the margins the transformed corners ask for, 2x up-sample, resample through G_inv, 2x down-sample and crop"""
        B, Cc, height, width = images.shape
        dev = images.device
        cx, cy = (width - 1) / 2, (height - 1) / 2
        cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float32, device=dev)
        cp = _mm(G_inv, cp.t().unsqueeze(0).expand(B, 3, 4))
        Hz_pad = self.Hz_geom.shape[0] // 4
        margin = cp[:, :2, :].permute(1, 0, 2).flatten(1)
        margin = torch.cat([-margin, margin]).max(dim=1).values
        margin = margin + torch.tensor([Hz_pad * 2 - cx, Hz_pad * 2 - cy] * 2, dtype=torch.float32, device=dev)
        margin = margin.max(torch.zeros(4, device=dev)).min(torch.tensor([width - 1, height - 1] * 2, dtype=torch.float32, device=dev))
        mx0, my0, mx1, my1 = (int(v) for v in margin.ceil().to(torch.int32).tolist())        # (one host round trip, as in the reference: the pad sizes are shapes)
        images = F.ReflectPad2dFn.apply(images, mx0, mx1, my0, my1)
        one = torch.ones([B], device=dev)
        G_inv = _mm(_translate2d((mx0 - mx1) / 2, (my0 - my1) / 2, one), G_inv)
        images = upfirdn2d.upsample2d(x=images, f=self.Hz_geom, up=2)
        G_inv = _mm(_mm(_scale2d(2, 2, one), G_inv), _scale2d(1 / 2, 1 / 2, one))
        G_inv = _mm(_mm(_translate2d(-0.5, -0.5, one), G_inv), _translate2d(0.5, 0.5, one))
        Ho, Wo = (height + Hz_pad * 2) * 2, (width + Hz_pad * 2) * 2
        G_inv = _mm(_mm(_scale2d(2 / images.shape[3], 2 / images.shape[2], one), G_inv), _scale2d(1 / (2 / Wo), 1 / (2 / Ho), one))
        images = F.AffineSampleFn.apply(images, G_inv[:, :2, :], Ho, Wo)
        return upfirdn2d.downsample2d(x=images, f=self.Hz_geom, down=2, padding=-Hz_pad * 2, flip_filter=True)

    def forward(self, images):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        L.require_gpu(images.device)
        B, Cc, height, width = images.shape
        dev = images.device
        G_inv = self._geometry(B, width, height, dev)
        if G_inv is not None:
            images = self._warp(images, G_inv)
        Cm = self._colour(B, Cc, dev)
        if Cm is not None:
            if Cc == 3:
                M = Cm[:, :3, :]
            elif Cc == 1:
                M = Cm[:, :3, :].mean(dim=1, keepdim=True)             # ada_aug.py:343-345
                M = torch.cat([M[:, :, :3].sum(dim=2, keepdim=True), torch.zeros(B, 1, 2, device=dev), M[:, :, 3:]], dim=2).expand(B, 3, 4)
            else:
                raise ValueError("Image must be RGB (3 channels) or L (1 channel)")
            images = F.ColorAffineFn.apply(images, M.contiguous())
        if self.imgfilter > 0:                                         # ada_aug.py:352-389
            num_bands = self.Hz_fbank.shape[0]
            assert len(self.imgfilter_bands) == num_bands
            expected_power = torch.as_tensor(np.array([10, 1, 1, 1]) / 13, dtype=torch.float32, device=dev)
            g = torch.ones([B, num_bands], device=dev)
            for i, band_strength in enumerate(self.imgfilter_bands):
                t_i = torch.exp2(torch.randn([B], device=dev) * self.imgfilter_std)
                t_i = torch.where(torch.rand([B], device=dev) < self.imgfilter * self.p * band_strength, t_i, torch.ones_like(t_i))
                t = torch.ones([B, num_bands], device=dev)
                t[:, i] = t_i
                t = t / (expected_power * t.square()).sum(dim=-1, keepdims=True).sqrt()
                g = g * t
            Hz_prime = (g.unsqueeze(-1) * self.Hz_fbank.unsqueeze(0)).sum(1)      # g @ Hz_fbank: [B, taps]
            images = F.FirReflectFn.apply(F.FirReflectFn.apply(images, Hz_prime, 0), Hz_prime, 1)
        noise = sigma = cut = None
        if self.noise > 0:                                             # ada_aug.py:393-399
            sigma = torch.randn([B, 1, 1, 1], device=dev).abs() * self.noise_std
            sigma = torch.where(torch.rand([B, 1, 1, 1], device=dev) < self.noise * self.p, sigma, torch.zeros_like(sigma))
            noise = torch.randn([B, Cc, height, width], device=dev)
        if self.cutout > 0:                                            # ada_aug.py:402-416
            size = torch.full([B, 2, 1, 1, 1], self.cutout_size, device=dev)
            size = torch.where(torch.rand([B, 1, 1, 1, 1], device=dev) < self.cutout * self.p, size, torch.zeros_like(size))
            center = torch.rand([B, 2, 1, 1, 1], device=dev)
            cut = torch.cat([center.reshape(B, 2), size.reshape(B, 2)], dim=1)
        if noise is not None or cut is not None:
            images = F.NoiseCutoutFn.apply(images, noise, sigma, cut)
        return images
