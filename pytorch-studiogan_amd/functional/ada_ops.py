"""image-side operators of adaptive discriminator augmentation (csrc/ext/ada.hip)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

# ---------------------------------------------------------------------------------------------------------
# adaptive discriminator augmentation: image-side operators (csrc/ext/ada.hip)
# ---------------------------------------------------------------------------------------------------------
def _f32_image(t, what):
    if t.dim() != 4 or t.dtype != torch.float32:
        raise RuntimeError(what + ": an fp32 [N, C, H, W] image batch expected")
    return _c(t)


class ReflectPad2dFn(torch.autograd.Function):
    """F.pad(x, [l, r, t, b], mode='reflect') (reference src/utils/ada_aug.py:265); backward = fold of the mirrored margins (gather form), its adjoint the pad again"""

    @staticmethod
    def forward(ctx, x, l, r, t, b):
        x = _f32_image(x, "reflect_pad2d")
        N, Cc, H, W = x.shape
        ctx.m = (l, r, t, b)
        y = torch.empty((N, Cc, H + t + b, W + l + r), dtype=torch.float32, device=x.device)
        L.call("sg_reflect_pad2d_fwd", L.ptr(x), L.ptr(y), N * Cc, H, W, l, r, t, b, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return ReflectPad2dBwdFn.apply(dy, *ctx.m), None, None, None, None


class ReflectPad2dBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, l, r, t, b):
        dy = _f32_image(dy, "reflect_pad2d backward")
        N, Cc, Ho, Wo = dy.shape
        ctx.m = (l, r, t, b)
        dx = torch.empty((N, Cc, Ho - t - b, Wo - l - r), dtype=torch.float32, device=dy.device)
        L.call("sg_reflect_pad2d_bwd", L.ptr(dy), L.ptr(dx), N * Cc, Ho - t - b, Wo - l - r, l, r, t, b, L.stream())
        return dx

    @staticmethod
    def backward(ctx, ddx):
        return ReflectPad2dFn.apply(ddx, *ctx.m), None, None, None, None


class AffineSampleFn(torch.autograd.Function):
    """grid_sample(x, affine_grid(theta, [N, C, Ho, Wo], align_corners=False)) with bilinear interpolation and zero padding (reference
    src/utils/ada_aug.py:276-277) in one pass; theta [N, 2, 3] is a draw (no gradient). Linear in x: backward and its adjoint are the two kernels."""

    @staticmethod
    def forward(ctx, x, theta, Ho, Wo):
        x = _f32_image(x, "affine_sample")
        theta = _c(theta.detach().float())
        N, Cc, Hi, Wi = x.shape
        if tuple(theta.shape) != (N, 2, 3):
            raise RuntimeError("affine_sample: theta must be [N, 2, 3]")
        ctx.save_for_backward(theta)
        ctx.dims = (Hi, Wi, Ho, Wo)
        y = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
        L.call("sg_affine_sample_fwd", L.ptr(x), L.ptr(theta), L.ptr(y), N, Cc, Hi, Wi, Ho, Wo, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        (theta,) = ctx.saved_tensors
        return AffineSampleBwdFn.apply(dy, theta, *ctx.dims), None, None, None


class AffineSampleBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, theta, Hi, Wi, Ho, Wo):
        dy = _f32_image(dy, "affine_sample backward")
        N, Cc = dy.shape[0], dy.shape[1]
        ctx.save_for_backward(theta)
        ctx.dims = (Ho, Wo)
        dx = torch.empty((N, Cc, Hi, Wi), dtype=torch.float32, device=dy.device)
        L.call("sg_affine_sample_bwd", L.ptr(dy), L.ptr(theta), L.ptr(dx), N, Cc, Hi, Wi, Ho, Wo, L.stream())
        return dx

    @staticmethod
    def backward(ctx, ddx):
        (theta,) = ctx.saved_tensors
        return AffineSampleFn.apply(ddx, theta, *ctx.dims), None, None, None, None, None


class ColorAffineFn(torch.autograd.Function):
    """y = M[:, :, :3] x + M[:, :, 3] per image (M [N, 3, 4]; one-plane images: y = x * M[n, 0, 0] + M[n, 0, 3]); reference src/utils/ada_aug.py:339-347.
    linear=True: without the offset column (the map applied to a cotangent)."""

    @staticmethod
    def forward(ctx, x, M, transpose=False):
        x = _f32_image(x, "color_affine")
        M = _c(M.detach().float())
        N, Cc, H, W = x.shape
        if tuple(M.shape) != (N, 3, 4):
            raise RuntimeError("color_affine: M must be [N, 3, 4]")
        ctx.save_for_backward(M)
        ctx.transpose = transpose
        y = torch.empty_like(x)
        L.call("sg_color_affine", L.ptr(x), L.ptr(M), L.ptr(y), N, Cc, H * W, 1 if transpose else 0, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        (M,) = ctx.saved_tensors
        if ctx.transpose:      # the adjoint of the adjoint: the linear part again (offset-free)
            M0 = M.clone()
            M0[:, :, 3] = 0
            return ColorAffineFn.apply(dy, M0, False), None, None
        return ColorAffineFn.apply(dy, M, True), None, None


class FirReflectFn(torch.autograd.Function):
    """one axis of ADA's per-image separable amplification filter over the reflect-padded image (reference src/utils/ada_aug.py:383-388: F.pad(mode='reflect') +
    grouped conv2d with one filter per image); taps [N, T] are derived from draws (no gradient). transpose=True: the adjoint; each is the other's backward."""

    @staticmethod
    def forward(ctx, x, taps, axis, transpose=False):
        x = _f32_image(x, "fir_reflect")
        taps = _c(taps.detach().float())
        N, Cc, H, W = x.shape
        if taps.dim() != 2 or taps.shape[0] != N:
            raise RuntimeError("fir_reflect: taps must be [N, T]")
        ctx.save_for_backward(taps)
        ctx.axis, ctx.transpose = axis, transpose
        y = torch.empty_like(x)
        L.call("sg_fir_reflect", L.ptr(x), L.ptr(taps), L.ptr(y), N, Cc, H, W, taps.shape[1], axis, 1 if transpose else 0, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        (taps,) = ctx.saved_tensors
        return FirReflectFn.apply(dy, taps, ctx.axis, not ctx.transpose), None, None, None


class NoiseCutoutFn(torch.autograd.Function):
    """y = (x + noise * sigma[n]) * cutout mask (reference src/utils/ada_aug.py:393-416); noise [N,C,H,W] / sigma [N] and cut [N,4] are draws. Linear in x up to the
    noise term: the backward is the mask alone."""

    @staticmethod
    def forward(ctx, x, noise, sigma, cut):
        x = _f32_image(x, "noise_cutout")
        N, Cc, H, W = x.shape
        noise = _c(noise.detach().float()) if noise is not None else None
        sigma = _c(sigma.detach().float().reshape(N)) if sigma is not None else None
        cut = _c(cut.detach().float().reshape(N, 4)) if cut is not None else None
        ctx.save_for_backward(cut)
        y = torch.empty_like(x)
        L.call("sg_ada_noise_cutout", L.ptr(x), L.ptr(noise), L.ptr(sigma), L.ptr(cut), L.ptr(y), N, Cc, H, W, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        (cut,) = ctx.saved_tensors
        if cut is None:
            return dy, None, None, None
        return NoiseCutoutFn.apply(dy, None, None, cut), None, None, None


__all__ = ['AffineSampleBwdFn', 'AffineSampleFn', 'ColorAffineFn', 'FirReflectFn', 'NoiseCutoutFn', 'ReflectPad2dBwdFn', 'ReflectPad2dFn', '_f32_image']
