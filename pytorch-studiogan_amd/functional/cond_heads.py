"""class-conditioning heads and losses (csrc/heads.hip; reference src/utils/losses.py:40-165)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

# ---------------------------------------------------------------------------------------------------------
# class-conditioning heads / losses (csrc/heads.hip; reference src/utils/losses.py:40-165,242-252)
# ---------------------------------------------------------------------------------------------------------
class RowNormalizeFn(torch.autograd.Function):
    """torch.nn.functional.normalize(x, dim=1, eps) for [B, d] fp32."""

    @staticmethod
    def forward(ctx, x, eps=1e-12):
        x = _c(x.float())
        y = torch.empty_like(x)
        inv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        L.call("sg_row_normalize_fwd", L.ptr(x), L.ptr(y), L.ptr(inv), x.shape[0], x.shape[1], float(eps), L.stream())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("RowNormalizeFn")
        y, inv = ctx.saved_tensors
        dy = _c(dy.float())
        dx = torch.empty_like(y)
        L.call("sg_row_normalize_bwd", L.ptr(y), L.ptr(inv), L.ptr(dy), L.ptr(dx), y.shape[0], y.shape[1], L.stream())
        return dx, None


class MatmulNTFn(torch.autograd.Function):
    """a [M, K] @ b [N, K]^T -> [M, N] in exact fp32 on the MFMA engine (similarity matrices of the contrastive losses)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a.float()), _c(b.float())
        M, K = a.shape
        N = b.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        gemm_raw(L.F32, b, 0, K, a, 0, K, out, N, N, M, K)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        _first_order_only("MatmulNTFn")
        a, b = ctx.saved_tensors
        g = _c(g.float())
        M, K = a.shape
        N = b.shape[0]
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)     # da[m][k] = sum_n g[m][n] b[n][k]
            gemm_raw(L.F32, b, 1, K, g, 0, N, da, K, K, M, N)
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)     # db[n][k] = sum_m g[m][n] a[m][k]
            gemm_raw(L.F32, a, 1, K, g, 1, N, db, K, K, N, M)
        return da, db


class RowDotFn(torch.autograd.Function):
    """p[r] = <a[r], b[r]>."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a.float()), _c(b.float())
        p = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        L.call("sg_row_dot", L.ptr(a), L.ptr(b), L.ptr(p), a.shape[0], a.shape[1], L.stream())
        ctx.save_for_backward(a, b)
        return p

    @staticmethod
    def backward(ctx, g):
        _first_order_only("RowDotFn")
        a, b = ctx.saved_tensors
        g = _c(g.float())
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            L.call("sg_row_scale", L.ptr(g), L.ptr(b), L.ptr(da), a.shape[0], a.shape[1], 0, L.stream())
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            L.call("sg_row_scale", L.ptr(g), L.ptr(a), L.ptr(db), a.shape[0], a.shape[1], 0, L.stream())
        return da, db


class ClassLossFn(torch.autograd.Function):
    """kind 0: mean cross entropy (torch.nn.CrossEntropyLoss, reference losses.py:40-47); 1: Crammer-Singer multi-hinge (losses.py:242-252)."""

    @staticmethod
    def forward(ctx, z, label, kind):
        z = _c(z.float())
        label = _c(label.long())
        rows, cols = z.shape
        row_loss = torch.empty(rows, dtype=torch.float32, device=z.device)
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        L.call("sg_class_loss", kind, L.ptr(z), L.ptr(label), rows, cols, L.ptr(row_loss), L.ptr(loss), L.ptr(dz), L.stream())
        ctx.save_for_backward(dz)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None


class ContrastiveLossFn(torch.autograd.Function):
    """kind 0: conditional contrastive loss (ContraGAN, losses.py:50-97); 1: data-to-data cross entropy (ReACGAN, losses.py:100-165) over the
    cosine-similarity matrix S [B, B] and the sample-to-proxy cosines p [B]."""

    @staticmethod
    def forward(ctx, S, p, label, kind, temperature, m_p):
        S, p, label = _c(S.float()), _c(p.float()), _c(label.long())
        B = S.shape[0]
        row_loss = torch.empty(B, dtype=torch.float32, device=S.device)
        loss = torch.empty(1, dtype=torch.float32, device=S.device)
        dS, dp = torch.empty_like(S), torch.empty_like(p)
        L.call("sg_contrastive_loss", kind, L.ptr(S), L.ptr(p), L.ptr(label), B, float(temperature), float(m_p), L.ptr(row_loss), L.ptr(loss),
               L.ptr(dS), L.ptr(dp), L.stream())
        ctx.save_for_backward(dS, dp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dS, dp = ctx.saved_tensors
        return dS * g, dp * g, None, None, None, None


class GatherColsFn(torch.autograd.Function):
    """z[r, label[r]] (multi-discriminator head, reference big_resnet.py:395-397)."""

    @staticmethod
    def forward(ctx, z, label):
        z, label = _c(z.float()), _c(label.long())
        out = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
        L.call("sg_gather_cols", L.ptr(z), L.ptr(label), z.shape[0], z.shape[1], L.ptr(out), L.stream())
        ctx.save_for_backward(label)
        ctx.shape = tuple(z.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (label,) = ctx.saved_tensors
        g = _c(g.float())
        dz = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        L.call("sg_scatter_cols", L.ptr(g), L.ptr(label), ctx.shape[0], ctx.shape[1], L.ptr(dz), L.stream())
        return dz, None


__all__ = ['ClassLossFn', 'ContrastiveLossFn', 'GatherColsFn', 'MatmulNTFn', 'RowDotFn', 'RowNormalizeFn']
