"""self-attention core (csrc/attn.hip; reference src/utils/ops.py:83-103)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)
from .layout import ConvertFn
from .conv import ConvCfg, ConvDgradFn

# ---------------------------------------------------------------------------------------------------------
# self-attention core (reference src/utils/ops.py:83-103)
# ---------------------------------------------------------------------------------------------------------
class MaxPool2Fn(torch.autograd.Function):
    """2x2 max pooling of an NHWC tensor -> [B, H/2 * W/2, C] (reference src/utils/ops.py:86,91: nn.MaxPool2d(2) on phi and g of SelfAttention). Its own autograd
    node since round 5 (the same two launches as before, when it sat inside the attention core's node) so that a create_graph pass can differentiate it: the
    backward is the scatter by the saved argmax, linear in dy, whose adjoint is the gather by the same argmax."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, H, W, Cc = x.shape
        y = torch.empty((B, (H // 2) * (W // 2), Cc), dtype=x.dtype, device=x.device)
        idx = torch.empty((B, (H // 2) * (W // 2), Cc), dtype=torch.uint8, device=x.device)
        L.call("sg_maxpool2_fwd", L.dt(x), L.ptr(x), Cc, L.ptr(y), Cc, L.ptr(idx), B, H, W, Cc, L.stream())
        ctx.save_for_backward(idx)
        ctx.dims = (B, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return MaxPool2BwdFn.apply(dy, idx, ctx.dims)
        return _maxpool2_bwd(dy, idx, ctx.dims)


def _maxpool2_bwd(dy, idx, dims):
    B, H, W, Cc = dims
    dy = _c(dy)
    dx = torch.empty((B, H, W, Cc), dtype=dy.dtype, device=dy.device)
    L.call("sg_maxpool2_bwd", L.dt(dy), L.ptr(dy), Cc, L.ptr(idx), L.ptr(dx), Cc, B, H, W, Cc, L.stream())
    return dx


class MaxPool2BwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, idx, dims):
        ctx.save_for_backward(idx)
        ctx.dims = dims
        return _maxpool2_bwd(dy, idx, dims)

    @staticmethod
    def backward(ctx, ddx):
        (idx,) = ctx.saved_tensors
        B, H, W, Cc = ctx.dims
        ddx = _c(ddx)
        y = torch.empty((B, (H // 2) * (W // 2), Cc), dtype=ddx.dtype, device=ddx.device)
        L.call("sg_maxpool2_gather", L.dt(ddx), L.ptr(ddx), Cc, L.ptr(idx), L.ptr(y), Cc, B, H, W, Cc, L.stream())
        return y, None, None


class BmmFn(torch.autograd.Function):
    """C[b] = op(A[b]) op(B[b]) in exact fp32 on the MFMA engine (op = transpose when ta / tb); closed under differentiation (its gradients are BmmFn calls):
    the matrix products of the create_graph pass through SelfAttention (reference src/utils/ops.py:93,100 torch.bmm, differentiated twice by autograd)."""

    @staticmethod
    def forward(ctx, A, Bm, ta, tb):
        A, Bm = _c(A), _c(Bm)
        if A.dtype != torch.float32 or Bm.dtype != torch.float32 or A.dim() != 3 or Bm.dim() != 3 or A.shape[0] != Bm.shape[0]:
            raise RuntimeError("BmmFn: two fp32 [batch, rows, cols] tensors expected")
        nb = A.shape[0]
        M, K = (A.shape[2], A.shape[1]) if ta else (A.shape[1], A.shape[2])
        K2, N = (Bm.shape[2], Bm.shape[1]) if tb else (Bm.shape[1], Bm.shape[2])
        if K != K2:
            raise RuntimeError("BmmFn: inner dimensions differ")
        out = torch.empty((nb, M, N), dtype=torch.float32, device=A.device)
        # OUT[j][i] = sum_k P(i, k) Q(j, k): P = op(B) seen from its column index (form 1 = [K][N] storage), Q = op(A) (form 0 = [M][K] storage)
        gemm_raw(L.F32, Bm, 0 if tb else 1, Bm.shape[2], A, 1 if ta else 0, A.shape[2], out, N, N, M, K, batch=nb,
                 p_bs=Bm.shape[1] * Bm.shape[2], q_bs=A.shape[1] * A.shape[2], out_bs=M * N)
        ctx.save_for_backward(A, Bm)
        ctx.t = (ta, tb)
        return out

    @staticmethod
    def backward(ctx, dC):
        A, Bm = ctx.saved_tensors
        ta, tb = ctx.t
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = BmmFn.apply(Bm, dC, tb, True) if ta else BmmFn.apply(dC, Bm, False, not tb)
        if ctx.needs_input_grad[1]:
            dB = BmmFn.apply(dC, A, True, ta) if tb else BmmFn.apply(A, dC, not ta, False)
        return dA, dB, None, None


class SoftmaxRowsFn(torch.autograd.Function):
    """P = softmax over the last dimension (fp32), differentiable twice: dS = P * (dP - <P, dP>) (SoftmaxRowsBwdFn), whose own gradients are the same map
    applied to the incoming cotangent (the Jacobian diag(P) - P P^T is symmetric) and sg_softmax_rows_bwd2 for the dependence on P."""

    @staticmethod
    def forward(ctx, S):
        S = _c(S)
        P = torch.empty_like(S)
        L.call("sg_softmax_rows", L.F32, L.ptr(S), L.ptr(P), S.numel() // S.shape[-1], S.shape[-1], L.stream())
        ctx.save_for_backward(P)
        return P

    @staticmethod
    def backward(ctx, dP):
        (P,) = ctx.saved_tensors
        return SoftmaxRowsBwdFn.apply(P, dP)


class SoftmaxRowsBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, dP):
        P, dP = _c(P), _c(dP)
        dS = torch.empty_like(P)
        L.call("sg_softmax_rows_bwd", L.F32, L.ptr(P), L.ptr(dP), L.ptr(dS), P.numel() // P.shape[-1], P.shape[-1], L.stream())
        ctx.save_for_backward(P, dP)
        return dS

    @staticmethod
    def backward(ctx, u):
        _first_order_only("SoftmaxRowsBwdFn")        # (third order: not a path of the reference's regularisers)
        P, dP = ctx.saved_tensors
        u = _c(u)
        rows, cols = P.numel() // P.shape[-1], P.shape[-1]
        gP = gdP = None
        if ctx.needs_input_grad[0]:
            gP = torch.empty_like(P)
            L.call("sg_softmax_rows_bwd2", L.ptr(P), L.ptr(dP), L.ptr(u), L.ptr(gP), rows, cols, L.stream())
        if ctx.needs_input_grad[1]:
            gdP = torch.empty_like(P)
            L.call("sg_softmax_rows_bwd", L.F32, L.ptr(P), L.ptr(u), L.ptr(gdP), rows, cols, L.stream())
        return gP, gdP


class ScalePtrFn(torch.autograd.Function):
    """y = sigma[0] * x with sigma a one-element fp32 device tensor (SelfAttention's learnt output gain, reference src/utils/ops.py:81,103) as a differentiable
    operator of both: dx = sigma * dy (itself again), dsigma = <dy, x> (accumulated into the parameter's gradient like every parameter gradient here)."""

    @staticmethod
    def forward(ctx, x, sigma):
        x = _c(x)
        y = torch.empty_like(x)
        L.call("sg_scale_by_ptr", L.dt(x), L.ptr(x), L.ptr(sigma), L.ptr(y), x.numel(), L.stream())
        ctx.save_for_backward(x, sigma)
        ctx.sigma_param = sigma
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sigma = ctx.saved_tensors
        dx = None
        if torch.is_grad_enabled():
            if _param_grad_wanted(ctx.sigma_param):
                raise NotImplementedError("create_graph=True is supported for input gradients only (WGAN-GP / R1 path)")
            return (ScalePtrFn.apply(dy, sigma) if ctx.needs_input_grad[0] else None), None
        dy = _c(dy)
        if ctx.needs_input_grad[1]:
            g = ensure_grad(ctx.sigma_param)
            L.call("sg_dot", L.dt(dy), L.ptr(dy), L.ptr(x), dy.numel(), L.ptr(g), 1.0, None, L.stream())
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            L.call("sg_scale_by_ptr", L.dt(dy), L.ptr(dy), L.ptr(sigma), L.ptr(dx), dy.numel(), L.stream())
        return dx, None


class AttnCoreFn:
    """o = softmax(theta . maxpool(phi)^T) . maxpool(g) per image; theta / phi: [B,H,W,Dp], g: [B,H,W,Cg] (reference src/utils/ops.py:83-101): the two poolings
    and the attention proper as three autograd nodes (the launches are those of the single node this replaced)."""

    @staticmethod
    def apply(theta, phi_full, g_full):
        return AttnPooledFn.apply(theta, MaxPool2Fn.apply(phi_full), MaxPool2Fn.apply(g_full))


def _attn_reference_graph(theta, phi, g, dims):
    """the attention core from differentiable fp32 primitives (scores and probabilities materialised): what a create_graph backward differentiates"""
    B, H, W, Dp, Cg = dims
    T = theta.dtype
    th = ConvertFn.apply(theta, torch.float32).reshape(B, H * W, Dp)
    ph, gg = ConvertFn.apply(phi, torch.float32), ConvertFn.apply(g, torch.float32)
    P = SoftmaxRowsFn.apply(BmmFn.apply(th, ph, False, True))
    o = BmmFn.apply(P, gg, False, False).reshape(B, H, W, Cg)
    return ConvertFn.apply(o, T)


class AttnPooledFn(torch.autograd.Function):
    """the attention core on pooled keys / values: theta [B,H,W,Dp], phi [B,HW/4,Dp], g [B,HW/4,Cg] -> o [B,H,W,Cg]"""

    @staticmethod
    def forward(ctx, theta, phi, g):
        theta, phi, g = _c(theta), _c(phi), _c(g)
        B, H, W, Dp = theta.shape
        Cg = g.shape[2]
        HW, HW4 = H * W, (H // 2) * (W // 2)
        dev, T = theta.device, theta.dtype
        sd = L.dt(T)
        fused = T == torch.bfloat16 and L.lib().sg_attn_fused_ok(B, HW, HW4, Dp, Cg) == 1
        # the bf16 probabilities are written only when a backward can come that needs them (they feed dg = P^T dO): never for the no-grad generator
        # forwards of the discriminator update, nor when the backward recomputes them itself (sg_attn_bwd_fused: no P and no dS in HBM at all)
        need_p = any(ctx.needs_input_grad) and L.lib().sg_attn_bwd_fused_ok(B, HW, HW4, Dp, Cg) != 1
        if need_p:
            fused_fwd = fused and L.lib().sg_attn_fwd_fused_ok(B, HW, HW4, Dp, Cg) == 1
        else:
            # keys and values streamed in 256-key chunks: no bound on the number of keys (16384 x 4096 scores per image in BigGAN-deep-256's D,
            # reference src/models/big_resnet_deep_legacy.py:80-95, never exist in HBM)
            fused_fwd = T == torch.bfloat16 and L.lib().sg_attn_fwd_flash_ok(B, HW, HW4, Dp, Cg) == 1
        lse = o32 = None
        if fused_fwd:
            # one launch: scores, softmax and the product with the pooled values
            P = torch.empty((B, HW, HW4), dtype=T, device=dev) if need_p else None
            lse = torch.empty((B, HW), dtype=torch.float32, device=dev)
            o = torch.empty((B, H, W, Cg), dtype=T, device=dev)
            # the fused backward takes delta_q = dO_q . O_q from an unrounded fp32 copy of the output instead of a pass over the keys
            o32 = torch.empty((B, HW, Cg), dtype=torch.float32, device=dev) if (P is None and any(ctx.needs_input_grad)) else None
            L.call("sg_attn_fwd_fused", L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(P), L.ptr(lse), L.ptr(o), L.ptr(o32), B, HW, HW4, Dp, Cg, L.stream())
        else:
            P = torch.empty((B, HW, HW4), dtype=T, device=dev)
            if fused:
                # scores stay in registers: one pass writes the bf16 probabilities (csrc/attn.hip)
                lse = torch.empty((B, HW), dtype=torch.float32, device=dev)
                L.call("sg_attn_probs_fwd", L.ptr(theta), L.ptr(phi), L.ptr(P), L.ptr(lse), B, HW, HW4, Dp, L.stream())
            else:
                S = torch.empty((B, HW, HW4), dtype=torch.float32, device=dev)
                # S[q][k] = theta_q . phi_k
                gemm_raw(sd, phi, 0, Dp, theta, 0, Dp, S, HW4, HW4, HW, Dp, batch=B, p_bs=HW4 * Dp, q_bs=HW * Dp, out_bs=HW * HW4, epi_flags=L.EPI_OUT_F32)
                L.call("sg_softmax_rows", sd, L.ptr(S), L.ptr(P), B * HW, HW4, L.stream())
                del S
            o = torch.empty((B, H, W, Cg), dtype=T, device=dev)
            # o[q][c] = sum_k P[q][k] g[k][c]
            gemm_raw(sd, g, 1, Cg, P, 0, HW4, o, Cg, Cg, HW, HW4, batch=B, p_bs=HW4 * Cg, q_bs=HW * HW4, out_bs=HW * Cg)
        ctx.save_for_backward(theta, phi, g, P, lse, o32)
        ctx.dims = (B, H, W, Dp, Cg)
        return o

    @staticmethod
    def backward(ctx, do):
        theta, phi, g, P, lse, o32 = ctx.saved_tensors
        B, H, W, Dp, Cg = ctx.dims
        if torch.is_grad_enabled():
            # create_graph=True (R1 / gradient penalties through a discriminator with attention): re-evaluate the block from differentiable primitives on the saved
            # inputs (which carry their graph) and let autograd take the first-order gradient of THAT with a graph of its own
            with torch.enable_grad():
                ins = [t for t, need in zip((theta, phi, g), ctx.needs_input_grad) if need]
                grads = list(torch.autograd.grad(_attn_reference_graph(theta, phi, g, ctx.dims), ins, do, create_graph=True)) if ins else []
            return tuple(grads.pop(0) if need else None for need in ctx.needs_input_grad)
        HW, HW4 = H * W, (H // 2) * (W // 2)
        do = _c(do)
        dev, T = do.device, do.dtype
        sd = L.dt(T)
        if P is None:
            # fused backward (csrc/attn.hip k_attn_bwd_q / k_attn_bwd_k): P, dP and dS are recomputed per tile in registers on both the
            # query side (dtheta) and the key side (dphi, dg); only the row statistics (lse, delta) cross HBM
            assert lse is not None
            delta = torch.empty((B, HW), dtype=torch.float32, device=dev)
            dtheta = torch.empty((B, H, W, Dp), dtype=T, device=dev)
            dphi = torch.empty((B, HW4, Dp), dtype=T, device=dev)
            dg = torch.empty((B, HW4, Cg), dtype=T, device=dev)
            L.call("sg_attn_bwd_fused", L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(do), L.ptr(o32), L.ptr(lse), L.ptr(delta), L.ptr(dtheta), L.ptr(dphi),
                   L.ptr(dg), B, HW, HW4, Dp, Cg, L.stream())
            return dtheta, dphi, dg
        # dg[k][c] = sum_q P[q][k] do[q][c]
        dg = torch.empty((B, HW4, Cg), dtype=T, device=dev)
        gemm_raw(sd, do, 1, Cg, P, 1, HW4, dg, Cg, Cg, HW4, HW, batch=B, p_bs=HW * Cg, q_bs=HW * HW4, out_bs=HW4 * Cg)
        dS = torch.empty((B, HW, HW4), dtype=T, device=dev)
        if lse is not None:
            # dS = P * (dP - sum_k P dP) with P and dP = dO . g^T recomputed in registers: no fp32 dP, no re-read of P (csrc/attn.hip)
            L.call("sg_attn_ds_bwd", L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(do), L.ptr(lse), L.ptr(dS), B, HW, HW4, Dp, Cg, L.stream())
        else:
            # dP[q][k] = sum_c do[q][c] g[k][c]
            dP = torch.empty((B, HW, HW4), dtype=torch.float32, device=dev)
            gemm_raw(sd, g, 0, Cg, do, 0, Cg, dP, HW4, HW4, HW, Cg, batch=B, p_bs=HW4 * Cg, q_bs=HW * Cg, out_bs=HW * HW4, epi_flags=L.EPI_OUT_F32)
            L.call("sg_softmax_rows_bwd", sd, L.ptr(P), L.ptr(dP), L.ptr(dS), B * HW, HW4, L.stream())
            del dP
        # dtheta[q][d] = sum_k dS[q][k] phi[k][d]
        dtheta = torch.empty((B, H, W, Dp), dtype=T, device=dev)
        gemm_raw(sd, phi, 1, Dp, dS, 0, HW4, dtheta, Dp, Dp, HW, HW4, batch=B, p_bs=HW4 * Dp, q_bs=HW * HW4, out_bs=HW * Dp)
        # dphi[k][d] = sum_q dS[q][k] theta[q][d]
        dphi = torch.empty((B, HW4, Dp), dtype=T, device=dev)
        gemm_raw(sd, theta, 1, Dp, dS, 1, HW4, dphi, Dp, Dp, HW4, HW, batch=B, p_bs=HW * Dp, q_bs=HW * HW4, out_bs=HW4 * Dp)
        return dtheta, dphi, dg


class AttnOutFn(torch.autograd.Function):
    """y = x + sigma * conv1x1(o)   (reference src/utils/ops.py:102-103), sigma read from device memory."""

    @staticmethod
    def forward(ctx, x, o, weight, sigma, rt, slot, link=None):
        bank = rt.bank()
        x, o = _c(x), _c(o)
        ctx.link = link
        y = conv2d_raw(o, bank.w_fwd(slot, rt), rt.Cin, rt.rows, 1, 1, res=x, alpha_ptr=sigma)
        ctx.save_for_backward(o, sigma)
        ctx.rt, ctx.slot = rt, slot
        ctx.sigma_param = sigma
        ctx.weight = weight      # the master parameter: only handed on to ConvDgradFn so the second-order graph reaches it
        return y

    @staticmethod
    def backward(ctx, dy):
        o, sigma = ctx.saved_tensors
        rt, slot = ctx.rt, ctx.slot
        if torch.is_grad_enabled():
            # create_graph=True: dx = dy and do = sigma * F_W^T(dy) as differentiable operators of dy (and, through ConvDgradFn / ScalePtrFn, of W and sigma)
            if _param_grad_wanted(ctx.weight, ctx.sigma_param):
                raise NotImplementedError("create_graph=True is supported for input gradients only (WGAN-GP / R1 path)")
            do = ScalePtrFn.apply(ConvDgradFn.apply(dy, o, ctx.weight, rt, slot, ConvCfg(1, 1)), ctx.sigma_param) if ctx.needs_input_grad[1] else None
            return (dy if ctx.needs_input_grad[0] else None), do, None, None, None, None, None
        bank = rt.bank()
        dy = _c(dy)
        N, H, W, Cc = dy.shape
        do = None
        if ctx.needs_input_grad[3]:
            t = conv2d_raw(o, bank.w_fwd(slot, rt), rt.Cin, rt.rows, 1, 1)  # recompute conv1x1(o)
            g = ensure_grad(ctx.sigma_param)
            L.call("sg_dot", L.dt(dy), L.ptr(dy), L.ptr(t), dy.numel(), L.ptr(g), 1.0, None, L.stream())
        if ctx.needs_input_grad[1]:
            do = conv2d_raw(dy, bank.w_dgrad(slot, rt), rt.rows, rt.Cin, 1, 1, alpha_ptr=sigma)
        if ctx.needs_input_grad[2]:
            conv2d_wgrad_raw(o, dy, bank.dwt(slot, rt), rt.Cin, rt.rows, 1, 1, H, W, alpha_ptr=sigma)
        dx = dy
        if ctx.link is not None and ctx.link.pending > 0 and ctx.needs_input_grad[0]:
            ctx.link.dx, dx = dy, None      # theta / phi / g read the same x: their data gradients pick this up as a residual (GradLink chain)
        return dx, do, None, None, None, None, None


__all__ = ['AttnCoreFn', 'AttnOutFn', 'AttnPooledFn', 'BmmFn', 'MaxPool2BwdFn', 'MaxPool2Fn', 'ScalePtrFn', 'SoftmaxRowsBwdFn', 'SoftmaxRowsFn', '_attn_reference_graph', '_maxpool2_bwd']
