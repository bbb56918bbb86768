"""differentiable augmentations in front of the discriminator and the consistency losses (csrc/ext/augment.hip, losses.hip, regularisers.hip)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

# ---------------------------------------------------------------------------------------------------------
# differentiable augmentations in front of the discriminator (csrc/ext/augment.hip)
# ---------------------------------------------------------------------------------------------------------
class AugSpec:
    """One sg_augment call: operator bits (applied in the kernel's fixed order), the per-image draws and the window sizes."""

    __slots__ = ("ops", "color", "geom", "cut_h", "cut_w", "max_t")

    def __init__(self, ops, color=None, geom=None, cut_h=0, cut_w=0, max_t=0):
        self.ops, self.color, self.geom, self.cut_h, self.cut_w, self.max_t = ops, color, geom, cut_h, cut_w, max_t


def _augment_launch(entry, spec, t, ops):
    import ctypes as C
    if t.dim() != 4 or not 1 <= t.shape[1] <= 4:
        raise RuntimeError("augment: expected an image batch [N, C <= 4, H, W]")
    if t.dtype != torch.float32:
        raise RuntimeError("augment: images cross the generator / discriminator boundary in fp32 (got %s)" % t.dtype)
    t = _c(t)
    N, Cc, H, W = t.shape
    for name, tab, width, dt_ in (("color", spec.color, 3, torch.float32), ("geom", spec.geom, 5, torch.int32)):
        if tab is not None and (tab.dtype != dt_ or tuple(tab.shape) != (N, width) or not tab.is_contiguous()):
            raise RuntimeError("augment: the %s table must be a contiguous [%d, %d] %s tensor" % (name, N, width, dt_))
    out = torch.empty_like(t)
    d = L.AugDesc(N, Cc, H, W, ops, spec.cut_h, spec.cut_w, spec.max_t, L.ptr(spec.color), L.ptr(spec.geom))
    work = torch.empty(L.lib().sg_augment_work_floats(C.byref(d)), dtype=torch.float32, device=t.device) if ops & L.AUG_CONTRAST else None
    L.call(entry, C.byref(d), L.ptr(t), L.ptr(out), L.ptr(work), L.stream())
    return out


class AugmentFn(torch.autograd.Function):
    """y = cutout(translate(flip(contrast(saturation(brightness(x)))))) in one gather pass (sg_augment_fwd; reference src/utils/diffaug.py:47-95,
    src/utils/cr.py:24-48). linear=True drops the brightness offset: the map applied to a cotangent in a create_graph pass."""

    @staticmethod
    def forward(ctx, x, spec, linear=False):
        ctx.spec = spec
        return _augment_launch("sg_augment_fwd", spec, x, spec.ops & ~L.AUG_BRIGHTNESS if linear else spec.ops)

    @staticmethod
    def backward(ctx, dy):
        return AugmentBwdFn.apply(dy, ctx.spec), None, None


class AugmentBwdFn(torch.autograd.Function):
    """dx = A^T dy for the linear part A of AugmentFn (sg_augment_bwd: the transposed gather); its own backward is A again, so R1 / gradient penalties
    through an augmented batch (reference src/worker.py:276-278 with :410-412) differentiate twice."""

    @staticmethod
    def forward(ctx, dy, spec):
        ctx.spec = spec
        return _augment_launch("sg_augment_bwd", spec, dy, spec.ops)

    @staticmethod
    def backward(ctx, ddx):
        return AugmentFn.apply(ddx, ctx.spec, True), None


class MseFn(torch.autograd.Function):
    """torch.nn.MSELoss() of two fp32 tensors (the reference's l2_loss, src/worker.py:116): fixed-order sum forward, one elementwise launch backward."""

    @staticmethod
    def forward(ctx, a, b):
        if a.shape != b.shape:
            raise RuntimeError("l2_loss: shapes differ (%s vs %s)" % (tuple(a.shape), tuple(b.shape)))
        a, b = _c(a.float()), _c(b.float())
        work = torch.empty(L.lib().sg_mse_work_floats(), dtype=torch.float32, device=a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        L.call("sg_mse_fwd", L.ptr(a), L.ptr(b), a.numel(), L.ptr(work), L.ptr(loss), L.stream())
        ctx.save_for_backward(a, b)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        _first_order_only("MseFn")
        a, b = ctx.saved_tensors
        g = _c(g.float().reshape(1))
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            L.call("sg_mse_bwd", L.ptr(a), L.ptr(b), L.ptr(g), a.numel(), L.ptr(da), L.ptr(db), L.stream())
        return da, db


class FeatureMatchingFn(torch.autograd.Function):
    """mean_c |mean_b fake_h[b, c] - mean_b real_h[b, c]| (reference src/utils/losses.py:254-259); gradient w.r.t. fake_h only (the worker detaches
    the real features, src/worker.py:594)."""

    @staticmethod
    def forward(ctx, real_h, fake_h):
        if real_h.dim() != 2 or real_h.shape != fake_h.shape:
            raise RuntimeError("feature_matching_loss: expected two [B, C] feature tensors of one shape")
        real_h, fake_h = _c(real_h.detach().float()), _c(fake_h.float())
        B, Cc = fake_h.shape
        work = torch.empty(L.lib().sg_fm_work_floats(Cc), dtype=torch.float32, device=fake_h.device)
        loss = torch.empty(1, dtype=torch.float32, device=fake_h.device)
        df = torch.empty_like(fake_h)
        L.call("sg_fm_loss", L.ptr(real_h), L.ptr(fake_h), B, Cc, L.ptr(work), L.ptr(loss), L.ptr(df), L.stream())
        ctx.save_for_backward(df)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return None, df * g


def _select_rows_raw(f, a, b):
    N = a.shape[0]
    out = torch.empty_like(a)
    L.call("sg_select_rows", L.ptr(f), L.ptr(a), L.ptr(b), L.ptr(out), N, a.numel() // N, L.stream())
    return out


class SelectRowsFn(torch.autograd.Function):
    """out[n] = a[n] if flag[n] else b[n]; differentiable in b (the real batch: the reference's fake * flag + real * (1 - flag) keeps the graph to real_images, which
    R1 differentiates twice, src/worker.py:274,379-381). The backward is the same launch on (0, g) and is itself a SelectRowsFn, so create_graph passes run through it."""

    @staticmethod
    def forward(ctx, f, a, b):
        ctx.save_for_backward(f)
        return _select_rows_raw(f, a, b)

    @staticmethod
    def backward(ctx, g):
        (f,) = ctx.saved_tensors
        g = _c(g)
        return None, None, SelectRowsFn.apply(f, torch.zeros_like(g), g)


def select_rows(flag, a, b):
    """out[n] = a[n] if flag[n] else b[n] for fp32 tensors of one shape (adaptive pseudo augmentation, reference src/utils/apa_aug.py:14-21). a (the detached fake
    batch, src/worker.py:274) carries no gradient; b (the real batch) does when it requires one."""
    if a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("select_rows: two fp32 tensors of one shape expected")
    f = _c(flag.to(torch.uint8))
    if f.numel() != a.shape[0]:
        raise RuntimeError("select_rows: one flag per row expected")
    a = _c(a.detach())
    if b.requires_grad and torch.is_grad_enabled():
        return SelectRowsFn.apply(f, a, _c(b))
    return _select_rows_raw(f, a, _c(b.detach()))


def sign_count_(acc, logits):
    """acc[0] += sum sign(logits), acc[1] += len(logits) on the device (the ADA / APA heuristic's accumulator, reference src/worker.py:285-289)."""
    lg = _c(logits.detach().float().reshape(-1))
    L.call("sg_sign_count", L.ptr(lg), lg.numel(), L.ptr(acc), L.stream())
    return acc


__all__ = ['AugSpec', 'AugmentBwdFn', 'AugmentFn', 'FeatureMatchingFn', 'MseFn', 'SelectRowsFn', '_augment_launch', '_select_rows_raw', 'select_rows', 'sign_count_']
