"""discriminator head, adversarial losses, gradient penalties, top-k (csrc/heads.hip, elementwise.hip; reference src/utils/losses.py:197-361)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

# ---------------------------------------------------------------------------------------------------------
# discriminator head + losses
# ---------------------------------------------------------------------------------------------------------
class ReluSumFn(torch.autograd.Function):
    """h[b,c] = sum_hw relu(x[b,hw,c])  (reference src/models/big_resnet.py:359-360)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, H, W, Cc = x.shape
        h = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        L.call("sg_relu_sum_hw_fwd", L.dt(x), L.ptr(x), L.ptr(h), B, H * W, Cc, L.stream())
        ctx.save_for_backward(x)
        return h

    @staticmethod
    def backward(ctx, dh):
        (x,) = ctx.saved_tensors
        return ReluSumBwdFn.apply(dh, x) if torch.is_grad_enabled() else _relu_sum_bwd(dh, x)


def _relu_sum_bwd(dh, x):
    B, H, W, Cc = x.shape
    dh = _c(dh.float())
    dx = torch.empty_like(x)
    L.call("sg_relu_sum_hw_bwd", L.dt(x), L.ptr(x), L.ptr(dh), L.ptr(dx), B, H * W, Cc, L.stream())
    return dx


class ReluSumBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dh, x):
        ctx.save_for_backward(x)
        return _relu_sum_bwd(dh, x)

    @staticmethod
    def backward(ctx, ddx):
        (x,) = ctx.saved_tensors
        ddx = _c(ddx)
        B, H, W, Cc = x.shape
        g = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        L.call("sg_masked_sum_hw", L.dt(x), L.ptr(ddx), L.ptr(x), L.ptr(g), B, H * W, Cc, L.stream())
        return g, None


class PDHeadFn(torch.autograd.Function):
    """adv[b] = linear1(h)[b] + <embed_sn(y_b), h_b>  (projection discriminator, reference big_resnet.py:363,387)."""

    @staticmethod
    def forward(ctx, h, w1, b1, emb_w, labels, rt_lin, rt_emb, slot):
        bank = rt_lin.bank()
        h = _c(h)
        B, Cc = h.shape
        dev = h.device
        emb = None
        if rt_emb is not None:
            labels = _c(labels.long())
            emb = torch.empty((B, Cc), dtype=torch.float32, device=dev)
            L.call("sg_embedding_fwd", bank.w_f32(slot, rt_emb), L.ptr(labels), L.ptr(emb), B, Cc, rt_emb.rows, L.stream())
        adv = torch.empty(B, dtype=torch.float32, device=dev)
        L.call("sg_pd_head_fwd", L.ptr(h), bank.w_f32(slot, rt_lin), L.ptr(b1), L.ptr(emb), L.ptr(adv), B, Cc, L.stream())
        ctx.save_for_backward(h, emb, labels if rt_emb is not None else None)
        ctx.rts = (rt_lin, rt_emb, slot)
        ctx.b1 = b1
        ctx.w1, ctx.emb_w = w1, emb_w    # master parameters, handed on to PDHeadBwdFn in a create_graph pass
        return adv

    @staticmethod
    def backward(ctx, dadv):
        h, emb, labels = ctx.saved_tensors
        rt_lin, rt_emb, slot = ctx.rts
        bank = rt_lin.bank()
        if torch.is_grad_enabled():
            if _param_grad_wanted(ctx.w1, ctx.emb_w, ctx.b1):
                raise NotImplementedError("create_graph=True is supported for input gradients only (WGAN-GP path)")
            dh = PDHeadBwdFn.apply(dadv, ctx.w1, ctx.emb_w, emb, labels, rt_lin, rt_emb, slot) if ctx.needs_input_grad[0] else None
            return dh, None, None, None, None, None, None, None
        B, Cc = h.shape
        dadv = _c(dadv.float())
        dh = torch.empty_like(h)
        train_w = ctx.needs_input_grad[1]
        dw1_dummy = None if train_w else zeros_small(Cc, torch.float32, h.device)   # kept alive until after the launch
        dw1 = bank.dwt(slot, rt_lin) if train_w else L.ptr(dw1_dummy)
        db1 = L.ptr(ensure_grad(ctx.b1)) if (ctx.b1 is not None and train_w) else None
        demb = torch.empty_like(emb) if emb is not None else None
        L.call("sg_pd_head_bwd", L.ptr(h), bank.w_f32(slot, rt_lin), L.ptr(emb), L.ptr(dadv), L.ptr(dh), dw1, db1, L.ptr(demb), B, Cc, L.stream())
        if emb is not None and ctx.needs_input_grad[3]:
            L.call("sg_embedding_bwd", L.ptr(demb), L.ptr(labels), bank.dwt(slot, rt_emb), B, Cc, rt_emb.rows, L.stream())
        return dh, None, None, None, None, None, None, None


class PDHeadBwdFn(torch.autograd.Function):
    """dh[b] = dadv[b] * (w1 + emb[y_b]) as a differentiable op of (dadv, w1, embedding): the same two head kernels with
    the roles h := ddh (second-order pass of the gradient penalty)."""

    @staticmethod
    def forward(ctx, dadv, w1, emb_w, emb, labels, rt_lin, rt_emb, slot):
        bank = rt_lin.bank()
        dadv = _c(dadv.float())
        B = dadv.numel()
        Cc = rt_lin.cols
        dev = dadv.device
        dh = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        scratch = zeros_small(Cc + 1, torch.float32, dev)
        hz = zeros_small((B, Cc), torch.float32, dev)
        demb = torch.empty((B, Cc), dtype=torch.float32, device=dev) if emb is not None else None
        L.call("sg_pd_head_bwd", L.ptr(hz), bank.w_f32(slot, rt_lin), L.ptr(emb), L.ptr(dadv), L.ptr(dh), L.ptr(scratch), None, L.ptr(demb), B, Cc, L.stream())
        ctx.save_for_backward(dadv, emb, labels)
        ctx.rts = (rt_lin, rt_emb, slot)
        return dh

    @staticmethod
    def backward(ctx, ddh):
        dadv, emb, labels = ctx.saved_tensors
        rt_lin, rt_emb, slot = ctx.rts
        bank = rt_lin.bank()
        ddh = _c(ddh.float())
        B, Cc = ddh.shape
        dev = ddh.device
        g_dadv = None
        if ctx.needs_input_grad[0]:
            g_dadv = torch.empty(B, dtype=torch.float32, device=dev)
            L.call("sg_pd_head_fwd", L.ptr(ddh), bank.w_f32(slot, rt_lin), None, L.ptr(emb), L.ptr(g_dadv), B, Cc, L.stream())
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw1_dummy = None if ctx.needs_input_grad[1] else zeros_small(Cc, torch.float32, dev)   # kept alive until after the launch
            dw1 = bank.dwt(slot, rt_lin) if ctx.needs_input_grad[1] else L.ptr(dw1_dummy)
            demb = torch.empty((B, Cc), dtype=torch.float32, device=dev) if emb is not None else None
            scratch = torch.empty((B, Cc), dtype=torch.float32, device=dev)
            L.call("sg_pd_head_bwd", L.ptr(ddh), bank.w_f32(slot, rt_lin), L.ptr(emb), L.ptr(dadv), L.ptr(scratch), dw1, None, L.ptr(demb), B, Cc, L.stream())
            if emb is not None and ctx.needs_input_grad[2]:
                L.call("sg_embedding_bwd", L.ptr(demb), L.ptr(labels), bank.dwt(slot, rt_emb), B, Cc, rt_emb.rows, L.stream())
        return g_dadv, None, None, None, None, None, None, None


class GradPenaltyFn(torch.autograd.Function):
    """kind 0: mean_b (||grads[b]||_2 - 1)^2 (reference utils/losses.py:313-315, :332-334); 1: 0.5 mean_b ||grads[b]||^2 (R1,
    :358-360); 2: max_b ||grads[b]||^2 (maxGP, :350-351)."""

    @staticmethod
    def forward(ctx, grads, kind=0):
        grads = _c(grads.float())
        B = grads.shape[0]
        n = grads.numel() // B
        norms = torch.empty(B + 1, dtype=torch.float32, device=grads.device)
        loss = torch.empty(1, dtype=torch.float32, device=grads.device)
        L.call("sg_gp_fwd", kind, L.ptr(grads), B, n, L.ptr(norms), L.ptr(loss), L.stream())
        ctx.save_for_backward(grads, norms)
        ctx.kind = kind
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        grads, norms = ctx.saved_tensors
        B = grads.shape[0]
        g = _c(gout.float().reshape(1))
        d = torch.empty_like(grads)
        L.call("sg_gp_bwd", ctx.kind, L.ptr(grads), L.ptr(norms), L.ptr(g), L.ptr(d), B, grads.numel() // B, L.stream())
        return d, None


def interpolate_rows(real, fake, alpha):
    """alpha[b] * real[b] + (1 - alpha[b]) * fake[b]  (reference utils/losses.py:303-308); fp32 NCHW in and out."""
    real, fake, alpha = _c(real.float()), _c(fake.float()), _c(alpha.float().reshape(-1))
    out = torch.empty_like(real)
    B = real.shape[0]
    L.call("sg_interp_rows", L.ptr(real), L.ptr(fake), L.ptr(alpha), L.ptr(out), B, real.numel() // B, L.stream())
    return out


class LeCamFn(torch.autograd.Function):
    """mean relu(real - ema_fake)^2 + mean relu(ema_real - fake)^2 (reference src/utils/losses.py:262-265)."""

    @staticmethod
    def forward(ctx, real, fake, ema_real, ema_fake):
        real, fake = _c(real.float().reshape(-1)), _c(fake.float().reshape(-1))
        B = real.numel()
        loss = torch.empty(1, dtype=torch.float32, device=real.device)
        dr, df = torch.empty_like(real), torch.empty_like(fake)
        L.call("sg_lecam", L.ptr(real), L.ptr(fake), B, float(ema_real), float(ema_fake), L.ptr(loss), L.ptr(dr), L.ptr(df), L.stream())
        ctx.save_for_backward(dr, df)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dr, df = ctx.saved_tensors
        return dr * g, df * g, None, None


def u8_to_nhwc(x, dtype, cpad=8, flip=None):
    """uint8 [N,H,W,3] (HDF5 / in-memory dataset layout, reference src/data_util.py:102-142) -> normalised NHWC tensor of the compute
    dtype with `cpad` channels: ToTensor + Normalize(0.5, 0.5) (+ per-image horizontal flip) in one kernel, no fp32 NCHW image."""
    x = _c(x)
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[3] == 3, "expected uint8 [N,H,W,3]"
    N, H, W, _ = x.shape
    y = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
    fl = None if flip is None else _c(flip.to(torch.uint8))
    L.call("sg_u8_to_nhwc", L.dt(y), L.ptr(x), L.ptr(fl), L.ptr(y), N, H, W, cpad, L.stream())
    return y


class TopkFn(torch.autograd.Function):
    """torch.topk(logits, k).values on a [B] vector (reference src/worker.py:565-566); backward scatters to the selected logits."""

    @staticmethod
    def forward(ctx, x, k):
        x = _c(x.float().reshape(-1))
        n = x.numel()
        vals = torch.empty(k, dtype=torch.float32, device=x.device)
        idx = torch.empty(k, dtype=torch.int32, device=x.device)
        L.call("sg_topk_select", L.ptr(x), n, k, L.ptr(vals), L.ptr(idx), L.stream())
        ctx.save_for_backward(idx)
        ctx.n = n
        return vals

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _c(g.float())
        dx = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
        L.call("sg_topk_scatter", L.ptr(g), L.ptr(idx), idx.numel(), L.ptr(dx), ctx.n, L.stream())
        return dx, None


_LOSS_KIND = {"hinge": 0, "wasserstein": 1, "vanilla": 2}


class DLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real, fake, kind):
        real, fake = _c(real.float()), _c(fake.float())
        B = real.numel()
        loss = torch.empty(1, dtype=torch.float32, device=real.device)
        dr, df = torch.empty_like(real), torch.empty_like(fake)
        if kind == 3:      # least squares (csrc/ext/losses.hip)
            L.call("sg_loss_ls_d", L.ptr(real), L.ptr(fake), B, L.ptr(loss), L.ptr(dr), L.ptr(df), L.stream())
        else:
            L.call("sg_loss_d", kind, L.ptr(real), L.ptr(fake), B, L.ptr(loss), L.ptr(dr), L.ptr(df), L.stream())
        ctx.save_for_backward(dr, df)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dr, df = ctx.saved_tensors
        return dr * g, df * g, None


class GLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fake, kind):
        fake = _c(fake.float())
        loss = torch.empty(1, dtype=torch.float32, device=fake.device)
        df = torch.empty_like(fake)
        if kind == 3:
            L.call("sg_loss_ls_g", L.ptr(fake), fake.numel(), L.ptr(loss), L.ptr(df), L.stream())
        else:
            L.call("sg_loss_g", kind, L.ptr(fake), fake.numel(), L.ptr(loss), L.ptr(df), L.stream())
        ctx.save_for_backward(df)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return df * g, None


__all__ = ['DLossFn', 'GLossFn', 'GradPenaltyFn', 'LeCamFn', 'PDHeadBwdFn', 'PDHeadFn', 'ReluSumBwdFn', 'ReluSumFn', 'TopkFn', '_LOSS_KIND', '_relu_sum_bwd', 'interpolate_rows', 'u8_to_nhwc']
