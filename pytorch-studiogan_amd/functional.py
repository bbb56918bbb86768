"""autograd.Function wrappers: every forward/backward below is one or more launches of libsgamd.so kernels.

Internal activation layout is NHWC ([N,H,W,C] contiguous, fp32 or bf16). Weight operands come from the network's
WeightBank slot of the current forward (see bank.py). Gradients w.r.t. parameters are written by the kernels straight
into the gradient arena (``p.grad`` views) -- the Functions return None for parameter inputs on purpose.
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import comm as _comm
from .bank import ensure_grad


def _first_order_only(name):
    """Functions without a differentiable backward refuse create_graph=True instead of silently cutting the graph."""
    if torch.is_grad_enabled():
        raise NotImplementedError(name + ": second-order gradients (create_graph=True) are implemented for the discriminator's "
                                         "conv / BN / pooling / self-attention / head path only (gradient penalties, R1)")


def _param_grad_wanted(*params):
    """Inside a backward: does the running graph task actually want the gradient of any of these leaves? (ctx.needs_input_grad
    is static; autograd.grad(inputs=...) -- the gradient penalty -- only wants the image gradient.)"""
    for p in params:
        if p is None or not torch.is_tensor(p) or not p.requires_grad:
            continue
        try:
            if torch._C._will_engine_execute_node(torch.autograd.graph.get_gradient_edge(p).node):
                return True
        except Exception:
            return True
    return False


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# Batch-norm statistics taken in the producing convolution's epilogue (csrc/conv_v2.h sg_conv_epilogue `stats`): the convolution offers them,
# the batch norm that runs as the VERY NEXT operator on exactly that tensor takes them (its statistics pass over the activation is then one
# small reduction over per-tile sums). _SEQ counts convolution / batch-norm forwards; an offer is only good for the operator right behind it.
_SEQ = [0]
_STATS_OFFER = [None]      # (seq, data_ptr, shape, per-tile sums [rows][C][2], rows, C)
_BN_FUSED_STATS = [os.environ.get("SG_BN_FUSED_STATS", "1") != "0"]
_CBN_MERGED = [os.environ.get("SG_CBN_MERGED", "1") != "0"]      # gain + bias linears of a conditional batch norm as one GEMM (CbnAffineFn)


def _tick():
    _SEQ[0] += 1


def _offer_stats(out, st, rows, C):
    # (the tensor's version counter rides along: an in-place write to the convolution's output between the two operators -- noise injection, a hook --
    # invalidates the offer instead of handing the batch norm statistics of what the tensor no longer holds)
    _STATS_OFFER[0] = (_SEQ[0], out.data_ptr(), tuple(out.shape), st, rows, C, out._version)


def _take_stats(x):
    ent, _STATS_OFFER[0] = _STATS_OFFER[0], None
    if ent is None or ent[0] != _SEQ[0] - 1 or ent[1] != x.data_ptr() or ent[2] != tuple(x.shape) or ent[5] != x.shape[3] or ent[6] != x._version:
        return None
    return ent[3], ent[4]


# ---------------------------------------------------------------------------------------------------------
# raw launch helpers (also used directly by the kernel-level tests)
# ---------------------------------------------------------------------------------------------------------
class f32_mode:
    """with f32_mode("bf16x3"): the generic engine's fp32 forward convolutions (all-vector operands) run as three bf16 MFMAs per 16-wide k-tile on operands split
    into two bf16 terms in registers -- fp32 tensors in and out, fp32 accumulation, ~2^-16 relative per product, 5.3x the matrix-pipe rate of the exact fp32 MFMA
    (csrc/gemm_core.h SPLIT, sg_set_f32_mode). "exact" (the default everywhere) = v_mfma_f32_32x32x2_f32. Process-wide switch: restored on exit."""
    MODES = {"exact": 0, "bf16x3": 3}

    def __init__(self, mode):
        if mode not in self.MODES:
            raise ValueError(f"f32_mode: {mode!r} (one of {sorted(self.MODES)})")
        self.mode = self.MODES[mode]

    def __enter__(self):
        self.saved = L.lib().sg_get_f32_mode()
        L.call("sg_set_f32_mode", self.mode)
        return self

    def __exit__(self, *exc):
        L.call("sg_set_f32_mode", self.saved)
        return False


def conv2d_raw(x, w_ptr, Cin, Cout, R, S, stride=1, pad_h=0, pad_w=0, pix_flags=0, epi_flags=0, bias=None, res=None, mask=None,
               alpha=1.0, beta=1.0, alpha_ptr=None, out=None, ldx=None, transposed_out_hw=None, out_coff=0, x_coff=0, desc=None):
    """x: [N,Hs,Ws,ldx] NHWC; returns [N,Ho',Wo',Cout]. w_ptr -> [Cout][R*S*Cin] in x.dtype."""
    N, Hs, Ws = x.shape[0], x.shape[1], x.shape[2]
    ldx = x.shape[3] if ldx is None else ldx
    up = 2 if (pix_flags & L.PIX_UPSAMPLE) else 1
    Hin, Win = Hs * up, Ws * up
    if pix_flags & L.PIX_TRANSPOSED:
        Ho, Wo = transposed_out_hw
    else:
        Ho = (Hin + 2 * pad_h - R) // stride + 1
        Wo = (Win + 2 * pad_w - S) // stride + 1
    pool = bool(epi_flags & L.EPI_POOL)
    Hy, Wy = (Ho // 2, Wo // 2) if pool else (Ho, Wo)
    if out is None:
        odt = torch.float32 if (epi_flags & L.EPI_OUT_F32) else x.dtype
        out = torch.empty((N, Hy, Wy, Cout), dtype=odt, device=x.device)
    d = desc if desc is not None else L.ConvFwdDesc()
    d.dtype = L.dt(x)
    d.N, d.Hs, d.Ws, d.C, d.ldx = N, Hs, Ws, Cin, ldx
    d.Ho, d.Wo, d.Cout = Ho, Wo, Cout
    d.R, d.S, d.stride, d.pad_h, d.pad_w = R, S, stride, pad_h, pad_w
    d.pix_flags, d.epi_flags = pix_flags, epi_flags
    d.alpha, d.beta = alpha, beta
    d.x, d.w = L.ptr(x) + x_coff * x.element_size(), w_ptr     # x_coff: read a channel slice of a wider tensor (pitch ldx)
    d.bias = L.ptr(bias)
    d.res = L.ptr(res)
    d.mask = L.ptr(mask)
    d.out = L.ptr(out) + out_coff * out.element_size()   # out_coff: write into a channel slice of a wider (concat) tensor
    d.alpha_ptr = L.ptr(alpha_ptr)
    d.ldo = out.shape[-1]
    d.ldr = res.shape[-1] if res is not None else 0
    d.ldm = mask.shape[-1] if mask is not None else 0
    if desc is not None:        # the caller launches (conv2d_skip_raw)
        return out
    L.call("sg_conv2d_fwd", d, L.stream())
    return out


def conv2d_skip_raw(x, w_ptr, Cin, Cout, x2, w2_ptr, C2, x2_up=False, pix_flags=0, epi_flags=0, bias=None, bias2=None, alpha=1.0, dry=False, stats=False):
    """[pool]( conv3x3(x; w) + conv1x1(up2?(x2); w2) ) + bias + bias2 in ONE launch (include/sgamd.h sg_conv2d_fwd_skip): the residual block's
    skip convolution as extra K-slices of its last 3x3 launch. dry=True: only ask whether the fused kernel takes the problem.
    Returns the output tensor, or None when the problem is not eligible (the caller then runs the two launches)."""
    sk = L.ConvSkipDesc()
    # (dry: eligibility does not depend on the output pointer's value, only on its alignment / pitch -- a real allocation is made anyway)
    out = conv2d_raw(x, w_ptr, Cin, Cout, 3, 3, 1, 1, 1, pix_flags, epi_flags, bias=bias, alpha=alpha, desc=sk.main)
    sk.x2, sk.w2, sk.bias2 = L.ptr(x2), w2_ptr, L.ptr(bias2)
    sk.C2, sk.ldx2, sk.x2_up = C2, x2.shape[3], 1 if x2_up else 0
    if L.lib().sg_conv2d_fwd_skip_ok(L.C.byref(sk)) != 1:
        return None
    if dry:
        return out
    if stats and not (epi_flags & L.EPI_POOL):
        # per-tile batch-norm statistics of the result from the epilogue (consumed by BNFn through _offer_stats / _take_stats)
        rows = L.lib().sg_conv2d_fwd_skip_stat_rows(L.C.byref(sk))
        st = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
        sk.stats = st.data_ptr()
        L.call("sg_conv2d_fwd_skip", sk, L.stream())
        _offer_stats(out, st, rows, Cout)
        return out
    L.call("sg_conv2d_fwd_skip", sk, L.stream())
    return out


def conv2d_wgrad_raw(x, dy, dw_ptr, Cin, Cout, R, S, Ho, Wo, stride=1, pad_h=0, pad_w=0, x_flags=0, g_flags=0, alpha=1.0,
                     alpha_ptr=None, splits=0, no_tr=0, ldg=None, dy_coff=0, dbias=None):
    """dw += alpha * wgrad(x, dy). dbias (fp32 [Cout], optional): asks the launch to add the bias gradient (column sums of the stored dy)
    as well; returns True when it did (halo kernel), False when the caller still has to run sg_colsum."""
    d = L.ConvWgradDesc()
    d.dtype = L.dt(x)
    d.N = x.shape[0]
    d.xHs, d.xWs, d.C, d.ldx, d.x_flags = x.shape[1], x.shape[2], Cin, x.shape[3], x_flags
    d.gHs, d.gWs, d.Cout, d.ldg, d.g_flags = dy.shape[1], dy.shape[2], Cout, (dy.shape[3] if ldg is None else ldg), g_flags
    d.Ho, d.Wo = Ho, Wo
    d.R, d.S, d.stride, d.pad_h, d.pad_w = R, S, stride, pad_h, pad_w
    d.alpha = alpha
    d.x, d.dy, d.dw = L.ptr(x), L.ptr(dy) + dy_coff * dy.element_size(), dw_ptr
    d.alpha_ptr = L.ptr(alpha_ptr)
    d.splits, d.no_tr = splits, no_tr
    fused = False
    if dbias is not None:
        d.dbias = L.ptr(dbias)
        fused = L.lib().sg_conv2d_wgrad_fuses_bias(L.C.byref(d)) == 1
        if not fused:
            d.dbias = None
    sp, wf = L.C.c_int(0), L.C.c_longlong(0)
    L.call("sg_conv2d_wgrad_plan", d, L.C.byref(sp), L.C.byref(wf))
    work = None
    if wf.value > 0:
        work = torch.empty(wf.value, dtype=torch.float32, device=x.device)   # scratch of the deterministic two-stage split-K
        d.splits, d.work, d.work_floats = sp.value, work.data_ptr(), wf.value
    L.call("sg_conv2d_wgrad", d, L.stream())
    return fused


def quad_pack_raw(src_ptr, dst, mode, M, Cs):
    """quad filter image [M][16][Cs] (dst tensor) of the 3x3 image at src_ptr ([M][9][Cs], dst.dtype); mode: include/sgamd.h sg_quad_pack"""
    L.call("sg_quad_pack", L.dt(dst), mode, src_ptr, L.ptr(dst), M, Cs, L.stream())
    return dst


def conv2d_q_raw(x, wq_ptr, form, Cin, Cout, pix_flags=0, epi_flags=0, bias=None, res=None, mask=None, alpha=1.0, beta=1.0, dry=False,
                 x2=None, w2q_ptr=None, bias2=None, stats=False, x2_norelu=False):
    """The quad forms of a 3x3 / pad-1 convolution next to a 2x resampling (include/sgamd.h sg_conv2d_q). form Q_POOL: x [N,2Hl,2Wl,C] ->
    [N,Hl,Wl,Cout] = avgpool2(conv3x3(x)); form Q_UP: x [N,Hl,Wl,C] -> [N,2Hl,2Wl,Cout] = conv3x3(up2(x)). Returns None when not eligible."""
    N = x.shape[0]
    if form == L.Q_POOL:
        if x.shape[1] % 2 or x.shape[2] % 2:
            return None
        Hl, Wl = x.shape[1] // 2, x.shape[2] // 2
        oshape = (N, Hl, Wl, Cout)
    else:
        Hl, Wl = x.shape[1], x.shape[2]
        oshape = (N, 2 * Hl, 2 * Wl, Cout)
    if x.dtype != torch.bfloat16:
        return None
    d = L.ConvQDesc()
    d.dtype, d.form = L.dt(x), form
    d.N, d.Hl, d.Wl, d.C, d.ldx, d.Cout = N, Hl, Wl, Cin, x.shape[3], Cout
    d.pix_flags, d.epi_flags, d.alpha, d.beta = pix_flags, epi_flags, alpha, beta
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    d.x, d.wq, d.bias, d.res, d.mask, d.out = L.ptr(x), wq_ptr, L.ptr(bias), L.ptr(res), L.ptr(mask), L.ptr(out)
    d.ldo = Cout
    d.ldr = res.shape[-1] if res is not None else 0
    d.ldm = mask.shape[-1] if mask is not None else 0
    if x2 is not None:      # Q_POOL: the block's 1x1 skip convolution in the same launch (x2: fine tensor, w2q: its filter x 1/4)
        d.x2, d.w2q, d.bias2, d.C2, d.ldx2 = L.ptr(x2), w2q_ptr, L.ptr(bias2), x2.shape[3], x2.shape[3]
        d.x2_norelu = 1 if x2_norelu else 0
    if L.lib().sg_conv2d_q_ok(L.C.byref(d)) != 1:
        return None
    if not dry:
        st = None
        if stats:
            rows = L.lib().sg_conv2d_q_stat_rows(L.C.byref(d))
            st = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
            d.stats = st.data_ptr()
        L.call("sg_conv2d_q", d, L.stream())
        if st is not None:
            _offer_stats(out, st, rows, Cout)
    return out


def conv2d_q_wgrad_raw(x, dy, dw_ptr, form, Cin, Cout, x_flags=0, alpha=1.0, dbias=None, splits=0):
    """dw3x3 (fp32 [Cout][9][Cin] at dw_ptr) += weight gradient of the quad form (include/sgamd.h sg_conv2d_q_wgrad); dbias += column sums of dy.
    Returns False when the kernel does not take the problem (nothing launched)."""
    if x.dtype != torch.bfloat16:
        return False
    d = L.ConvQWgradDesc()
    d.dtype, d.form = L.dt(x), form
    lo = dy if form == L.Q_POOL else x
    d.N, d.Hl, d.Wl = lo.shape[0], lo.shape[1], lo.shape[2]
    d.C, d.ldx, d.x_flags, d.Cout, d.ldg = Cin, x.shape[3], x_flags, Cout, dy.shape[3]
    d.alpha = alpha
    d.x, d.dy, d.dw, d.dbias = L.ptr(x), L.ptr(dy), dw_ptr, L.ptr(dbias)
    d.splits = splits
    sp, wf = L.C.c_int(0), L.C.c_longlong(0)
    L.call("sg_conv2d_q_wgrad_plan", d, L.C.byref(sp), L.C.byref(wf))
    if sp.value == 0:
        return False
    work = torch.empty(wf.value, dtype=torch.float32, device=x.device)
    d.work, d.work_floats, d.splits = work.data_ptr(), wf.value, sp.value
    L.call("sg_conv2d_q_wgrad", d, L.stream())
    return True


def gemm_raw(dtype, p, p_form, ldp, q, q_form, ldq, out, ldo, I, J, K, batch=1, p_bs=0, q_bs=0, out_bs=0, bias=None, res=None,
             res_bs=0, ldr=0, beta=1.0, alpha=1.0, alpha_ptr=None, epi_flags=0, splits=1, no_tr=0):
    """OUT[b][j][i] = beta*res + alpha * sum_k P(i,k) Q(j,k) + bias[i]; p/q/out may be tensors or raw pointers."""
    d = L.GemmDesc()
    d.dtype, d.p_form, d.q_form = dtype, p_form, q_form
    d.I, d.J, d.K, d.batch = I, J, K, batch
    d.p = p if isinstance(p, int) else L.ptr(p)
    d.q = q if isinstance(q, int) else L.ptr(q)
    d.out = out if isinstance(out, int) else L.ptr(out)
    d.p_bstride, d.ldp, d.q_bstride, d.ldq, d.out_bstride, d.ldo = p_bs, ldp, q_bs, ldq, out_bs, ldo
    d.bias = bias if (bias is None or isinstance(bias, int)) else L.ptr(bias)
    d.res = res if (res is None or isinstance(res, int)) else L.ptr(res)
    d.res_bstride, d.ldr, d.beta = res_bs, ldr, beta
    d.alpha = alpha
    d.alpha_ptr = L.ptr(alpha_ptr)
    d.epi_flags, d.splits, d.no_tr = epi_flags, splits, no_tr
    L.call("sg_gemm", d, L.stream())


_DGRAD_SPLITK = [os.environ.get("SG_DGRAD_SPLITK", "1") != "0"]      # SG_DGRAD_SPLITK=0: one launch, one serial contraction per output tile (A/B runs)


def gemm_dgrad_rows(w_ptr, dy, dx, B, K, O):
    """dx[b][k] = sum_o dy[b][o] W[o][k] for a (sn)linear layer's [O][K] fp32 weight image (fp32): the contraction runs over the layer's OUTPUT width, the result
    is only [B][K]. At BigGAN's conditional batch norms (K = 148, B = 256, O = 2 C up to 3072) the plain launch is FOUR workgroups walking 3072 contraction steps each:
    270 us (tools/cbn_gemm_bench.py, profiles/r05_cbn_gemm_bench.txt; forward and weight gradient of the same layer: 23 / 30 us) -- 1.3 ms per generator backward.
    Here the contraction is cut into S slices that run as the S batches of ONE launch (batch strides = slice offsets along the contraction) into [S][B][K] partial
    results, summed in a fixed order by one small reduction: no atomics, bit-reproducible."""
    S = 1
    tiles = ((K + 127) // 128) * ((B + 127) // 128)          # output tiles of the plain launch
    if _DGRAD_SPLITK[0] and tiles < 64:
        while S < 32 and tiles * S < 256 and O % (2 * S) == 0 and O // (2 * S) >= 96:
            S *= 2
    if S == 1:
        gemm_raw(L.F32, w_ptr, 1, K, dy, 0, dy.shape[1] if torch.is_tensor(dy) else O, dx, K, K, B, O)
        return dx
    ldq = dy.shape[1]
    ws = torch.empty((S, B, K), dtype=torch.float32, device=dx.device)
    gemm_raw(L.F32, w_ptr, 1, K, dy, 0, ldq, ws, K, K, B, O // S, batch=S, p_bs=(O // S) * K, q_bs=O // S, out_bs=B * K)
    torch.sum(ws, dim=0, out=dx)
    return dx


# ---------------------------------------------------------------------------------------------------------
# layout at the reference's NCHW fp32 boundary
# ---------------------------------------------------------------------------------------------------------
class NchwToNhwcFn(torch.autograd.Function):
    """fp32 NCHW -> compute-dtype NHWC (D input image; G's linear0 output). cpad > C: the NHWC tensor gets cpad channels, the extra
    ones zero (RGB images travel as 8-channel tensors so every convolution uses the 16-byte loaders)."""

    @staticmethod
    def forward(ctx, x, dtype, cpad=0):
        x = _c(x)
        N, Cc, H, W = x.shape
        ld = max(cpad, Cc)
        y = (torch.zeros if ld > Cc else torch.empty)((N, H, W, ld), dtype=dtype, device=x.device)
        L.call("sg_nchw_to_nhwc", L.dt(dtype), L.ptr(x), L.ptr(y), N, Cc, H, W, ld, L.stream())
        ctx.channels = Cc
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():        # create_graph=True (WGAN-GP): stay on differentiable ops
            return NhwcToNchwFn.apply(dy, False, ctx.channels), None, None
        dy = _c(dy)
        N, H, W, ld = dy.shape
        Cc = ctx.channels
        dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dy.device)
        L.call("sg_nhwc_to_nchw", L.dt(dy), L.ptr(dy), L.ptr(dx), N, Cc, H, W, ld, 0, L.stream())
        return dx, None, None


class NhwcToNchwFn(torch.autograd.Function):
    """compute-dtype NHWC -> fp32 NCHW with optional tanh (G output image). channels < x.shape[3]: only the first `channels`
    are real (the last convolution of G writes 8-channel rows)."""

    @staticmethod
    def forward(ctx, x, apply_tanh, channels=0):
        x = _c(x)
        N, H, W, ld = x.shape
        Cc = channels or ld
        y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
        L.call("sg_nhwc_to_nchw", L.dt(x), L.ptr(x), L.ptr(y), N, Cc, H, W, ld, 1 if apply_tanh else 0, L.stream())
        ctx.apply_tanh = apply_tanh
        ctx.in_dtype = x.dtype
        ctx.ld = ld
        if apply_tanh:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():        # create_graph=True (latent optimisation: d D(G(z)) / dz is differentiated again): stay on differentiable operators
            t = TanhGradFn.apply(dy, ctx.saved_tensors[0]) if ctx.apply_tanh else dy
            return NchwToNhwcFn.apply(t, ctx.in_dtype, ctx.ld), None, None
        dy = _c(dy.float())
        N, Cc, H, W = dy.shape
        y = ctx.saved_tensors[0] if ctx.apply_tanh else None
        ld = ctx.ld
        dx = (torch.zeros if ld > Cc else torch.empty)((N, H, W, ld), dtype=ctx.in_dtype, device=dy.device)
        L.call("sg_nchw_grad_to_nhwc", L.dt(ctx.in_dtype), L.ptr(dy), L.ptr(y), L.ptr(dx), N, Cc, H, W, ld, 1 if ctx.apply_tanh else 0, L.stream())
        return dx, None, None


class TanhGradFn(torch.autograd.Function):
    """t = dy * (1 - y^2) with y = tanh(x) the forward's OUTPUT (so its gradient re-enters the producing node): the tanh backward as a differentiable operator"""

    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _c(dy.float()), _c(y.float())
        t = torch.empty_like(dy)
        L.call("sg_tanh_bwd", L.ptr(dy), L.ptr(y), L.ptr(t), dy.numel(), L.stream())
        ctx.save_for_backward(dy, y)
        return t

    @staticmethod
    def backward(ctx, g):
        _first_order_only("TanhGradFn")
        dy, y = ctx.saved_tensors
        g = _c(g.float())
        g_dy = g_y = None
        if ctx.needs_input_grad[0]:
            g_dy = torch.empty_like(g)
            L.call("sg_tanh_bwd", L.ptr(g), L.ptr(y), L.ptr(g_dy), g.numel(), L.stream())
        if ctx.needs_input_grad[1]:
            g_y = torch.empty_like(g)
            L.call("sg_tanh_bwd2", L.ptr(g), L.ptr(dy), L.ptr(y), L.ptr(g_y), g.numel(), L.stream())
        return g_dy, g_y


class ConvertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        x = _c(x)
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        L.call("sg_convert", L.dt(x), L.dt(dtype), L.ptr(x), L.ptr(y), x.numel(), L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():          # a create_graph pass: the conversion is linear, its adjoint is the conversion back
            return ConvertFn.apply(dy, ctx.src), None
        dy = _c(dy)
        dx = torch.empty(dy.shape, dtype=ctx.src, device=dy.device)
        L.call("sg_convert", L.dt(dy), L.dt(ctx.src), L.ptr(dy), L.ptr(dx), dy.numel(), L.stream())
        return dx, None


# ---------------------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------------------
class ConvCfg:
    __slots__ = ("R", "S", "stride", "pad_h", "pad_w", "in_relu", "in_upsample", "out_pool", "stats")

    def __init__(self, R, S, stride=1, pad_h=0, pad_w=0, in_relu=False, in_upsample=False, out_pool=False, stats=False):
        self.R, self.S, self.stride, self.pad_h, self.pad_w = R, S, stride, pad_h, pad_w
        self.in_relu, self.in_upsample, self.out_pool = in_relu, in_upsample, out_pool
        self.stats = stats      # a batch norm reads the result next: take its statistics in the epilogue where the kernel can


class GradLink:
    """Carries the gradient a residual block's INPUT receives through the skip path from the block tail's backward (ConvSkipFn) to the
    backward of the block's first operator (ConvFn of a discriminator block, BNFn of a generator block), which adds it in its own launch
    (sg_conv2d_fwd with mask AND residual / sg_bn_bwd_apply_res). Without it autograd sums the two contributions in a separate elementwise
    launch per block (343 `add<bf16>` launches, 3.3 ms per C3 step in profiles/r03_bench_biggan128_bs256_kerneltrace_a.txt).
    The tail's backward always runs first (autograd executes nodes in reverse creation order), stashes its dx here and returns None for
    that input; create_graph passes (gradient penalty) do not use the link.

    chain=True (SelfAttention): SEVERAL convolutions read the same x (theta / phi / g) next to the residual. AttnOutFn's backward stashes the
    residual's gradient; each of the convolutions takes what is stashed as the residual of its own data-gradient launch and stashes the sum
    again, and the one that runs last (`pending` counts them, so the order among them does not matter) hands the total to autograd: the three
    `add<bf16>` launches per attention backward (0.3 ms at 64 x 64 x 96, batch 256) become three residual reads."""
    __slots__ = ("dx", "chain", "pending")

    def __init__(self, chain=False):
        self.dx = None
        self.chain = chain
        self.pending = 0

    def take(self):
        t, self.dx = self.dx, None
        return t


_GRAD_LINK = [os.environ.get("SG_GRAD_LINK", "1") != "0"]      # SG_GRAD_LINK=0: leave the sum to autograd (A/B runs, tests)


_QUAD = [os.environ.get("SG_QUAD", "1") != "0"]      # SG_QUAD=0: the 3x3 kernels everywhere (A/B runs, tests)


def _quad_form(rt, cfg, x):
    """Q_POOL / Q_UP when this launch is a 3x3 / pad-1 convolution next to a 2x resampling that the quad kernels take (csrc/conv_q.h: the same
    result through the pooled / phase filters, 16 C instead of 36 C MACs per low-resolution position), else None."""
    if not _QUAD[0] or x.dtype != torch.bfloat16 or cfg.R != 3 or cfg.S != 3 or cfg.stride != 1 or cfg.pad_h != 1 or cfg.pad_w != 1:
        return None
    if cfg.out_pool == cfg.in_upsample or rt.trans or rt.RS != 9:
        return None
    if rt.cin_pad % 32 or (rt.rows_pad % 64 and rt.rows_pad % 96):
        return None
    return L.Q_POOL if cfg.out_pool else L.Q_UP


def _conv_fwd(x, rt, slot, cfg, bias, res=None, stats=False):
    """ConvFn's forward launch: [res +] avgpool2?(conv(up2?(relu?(x)))) + bias. stats: also offer the result's batch-norm statistics (quad kernel)"""
    bank = rt.bank()
    Cin = x.shape[3]
    form = _quad_form(rt, cfg, x)
    if form is not None:
        y = conv2d_q_raw(x, bank.w_quad(slot, rt, form), form, Cin, rt.rows_pad, L.PIX_RELU if cfg.in_relu else 0, 0, bias=bias, res=res,
                         stats=stats and _BN_FUSED_STATS[0] and rt.rows_pad == rt.rows)
        if y is not None:
            return y
    pf = (L.PIX_RELU if cfg.in_relu else 0) | (L.PIX_UPSAMPLE if cfg.in_upsample else 0)
    ef = L.EPI_POOL if cfg.out_pool else 0
    return conv2d_raw(x, bank.w_fwd(slot, rt), Cin, rt.rows_pad, cfg.R, cfg.S, cfg.stride, cfg.pad_h, cfg.pad_w, pf, ef, bias=bias, res=res,
                      alpha=0.25 if cfg.out_pool else 1.0)


def _conv_wgrad(x, dy, rt, slot, cfg, dbias=None):
    """weight gradient of ConvFn's launch into the bank's fp32 scratch; returns True when the bias gradient (dbias) was produced on the side"""
    bank = rt.bank()
    N, Hs, Ws, Cin = x.shape
    form = _quad_form(rt, cfg, x)
    if form is not None and conv2d_q_wgrad_raw(x, dy, bank.dwt(slot, rt), form, Cin, rt.rows_pad, L.PIX_RELU if cfg.in_relu else 0, dbias=dbias):
        return dbias is not None
    up = 2 if cfg.in_upsample else 1
    Ho = (Hs * up + 2 * cfg.pad_h - cfg.R) // cfg.stride + 1
    Wo = (Ws * up + 2 * cfg.pad_w - cfg.S) // cfg.stride + 1
    pool = cfg.out_pool
    xf = (L.PIX_RELU if cfg.in_relu else 0) | (L.PIX_UPSAMPLE if cfg.in_upsample else 0)
    return conv2d_wgrad_raw(x, dy, bank.dwt(slot, rt), Cin, rt.rows_pad, cfg.R, cfg.S, Ho, Wo, cfg.stride, cfg.pad_h, cfg.pad_w, xf,
                            L.PIX_UPSAMPLE if pool else 0, alpha=0.25 if pool else 1.0, dbias=dbias)


def _conv_dgrad(dy, x, rt, slot, cfg, res=None):
    """data gradient of ConvFn's fused launch: dx = relu-mask(x) * F^T(dy) [+ res], F = pool?(conv(up?(.))) * (0.25 if pool)."""
    bank = rt.bank()
    N, Hs, Ws, Cin = x.shape
    form = _quad_form(rt, cfg, x)
    if form is not None and dy.shape[3] == rt.rows_pad:
        # the data gradient of one quad form is the other form with the transformed flipped image (sg_quad_pack modes 2 / 3)
        dx = conv2d_q_raw(dy, bank.w_quad(slot, rt, 2 + form), 1 - form, rt.rows_pad, Cin, 0, 0, mask=x if cfg.in_relu else None, res=res)
        if dx is not None:
            return dx
    up = 2 if cfg.in_upsample else 1
    Hin, Win = Hs * up, Ws * up
    pool = cfg.out_pool
    if cfg.stride != 1:
        # strided convolution: gather form of the transposed convolution with the UNflipped [Cin][r][s][Cout] image
        assert not (pool or cfg.in_upsample), "upsample / pooling fusion is stride-1 only"
        return conv2d_raw(dy, bank.w_dgrad(slot, rt), rt.rows_pad, Cin, cfg.R, cfg.S, cfg.stride, cfg.pad_h, cfg.pad_w, L.PIX_TRANSPOSED, 0,
                          mask=x if cfg.in_relu else None, res=res, transposed_out_hw=(Hin, Win), ldx=dy.shape[3])
    pf = L.PIX_UPSAMPLE if pool else 0
    ef = L.EPI_POOL if cfg.in_upsample else 0
    # dy has rows_pad channels and the dgrad image is [cin_pad][R][S][rows_pad] (zero outside the real weights): the padded
    # channels ride along so the 16-byte loaders apply; dx comes out with cin_pad channels like x
    return conv2d_raw(dy, bank.w_dgrad(slot, rt), rt.rows_pad, Cin, cfg.R, cfg.S, 1, cfg.R - 1 - cfg.pad_h, cfg.S - 1 - cfg.pad_w, pf, ef,
                      mask=x if cfg.in_relu else None, res=res, alpha=0.25 if pool else 1.0, ldx=dy.shape[3])


class ConvDgradFn(torch.autograd.Function):
    """The data gradient of ConvFn as a differentiable op (second-order pass of the gradient penalty, reference
    utils/losses.py:301-316). dx = M * F_W^T(dy) is linear in dy and in the weight image, so with t = M * ddx:
        d/d(dy) = F_W(t)           -- the forward launch again, without bias / residual / ReLU-on-load
        d/dW    = wgrad(t, dy)     -- the forward's weight-gradient launch with x := t, accumulated into the bank's scratch
    (M, the ReLU mask of the saved input, is piecewise constant)."""

    @staticmethod
    def forward(ctx, dy, x, weight, rt, slot, cfg):
        dy = _c(dy)
        ctx.save_for_backward(dy, x)
        ctx.rt, ctx.slot, ctx.cfg = rt, slot, cfg
        return _conv_dgrad(dy, x, rt, slot, cfg)

    @staticmethod
    def backward(ctx, ddx):
        dy, x = ctx.saved_tensors
        rt, slot, cfg = ctx.rt, ctx.slot, ctx.cfg
        bank = rt.bank()
        t = _c(ddx)
        if cfg.in_relu:
            m = torch.empty_like(t)
            L.call("sg_relu_mask", L.dt(t), L.ptr(t), L.ptr(x), L.ptr(m), t.numel(), L.stream())
            t = m
        N, Hs, Ws, Cin = x.shape
        up = 2 if cfg.in_upsample else 1
        Ho = (Hs * up + 2 * cfg.pad_h - cfg.R) // cfg.stride + 1
        Wo = (Ws * up + 2 * cfg.pad_w - cfg.S) // cfg.stride + 1
        pool = cfg.out_pool
        g_dy = None
        if ctx.needs_input_grad[0]:
            # (rows_pad: the zero rows of a padded weight image -- theta / phi of SelfAttention, the RGB layer -- ride along as in the first-order launches)
            g_dy = conv2d_raw(t, bank.w_fwd(slot, rt), Cin, rt.rows_pad, cfg.R, cfg.S, cfg.stride, cfg.pad_h, cfg.pad_w,
                              L.PIX_UPSAMPLE if cfg.in_upsample else 0, L.EPI_POOL if pool else 0, alpha=0.25 if pool else 1.0)
        if ctx.needs_input_grad[2]:
            conv2d_wgrad_raw(t, dy, bank.dwt(slot, rt), Cin, rt.rows_pad, cfg.R, cfg.S, Ho, Wo, cfg.stride, cfg.pad_h, cfg.pad_w,
                             L.PIX_UPSAMPLE if cfg.in_upsample else 0, L.PIX_UPSAMPLE if pool else 0, alpha=0.25 if pool else 1.0)
        return g_dy, None, None, None, None, None


class ConvFn(torch.autograd.Function):
    """y = [res +] avgpool2?( conv( upsample2?( relu?(x) ) ) + bias )      (one fused implicit-GEMM launch)

    backward: data gradient = same engine with the flipped/transposed weight image, ReLU mask / 2x2 pooling-sum /
    pooled-gradient broadcast fused; weight gradient = split-K MFMA contraction over pixels into the bank's fp32 scratch;
    bias gradient = column sums.  Replaces nn.Conv2d fwd/bwd + ReLU + F.interpolate + AvgPool2d + add of
    reference src/models/big_resnet.py:28-42,177-242.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, res, rt, slot, cfg, link=None):
        bank = rt.bank()
        x = _c(x)
        N, Hs, Ws, Cin = x.shape
        assert Cin == rt.cin_pad, f"conv input channels {Cin} != {rt.cin_pad}"
        ctx.link = link
        if link is not None and link.chain:
            link.pending += 1
        if res is not None:
            res = _c(res)
        bias_k = bias
        if bias is not None and rt.rows_pad != rt.rows:      # padded output channels: the epilogue reads rows_pad bias entries
            bias_k = torch.zeros(rt.rows_pad, dtype=torch.float32, device=x.device)
            bias_k[:rt.rows].copy_(bias.detach())
        _tick()
        y = _conv_fwd(x, rt, slot, cfg, bias_k, res, stats=cfg.stats)
        ctx.save_for_backward(x)
        ctx.rt, ctx.slot, ctx.cfg = rt, slot, cfg
        ctx.bias = bias
        ctx.weight = weight      # the master parameter: only handed on to ConvDgradFn so the second-order graph reaches it
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        rt, slot, cfg = ctx.rt, ctx.slot, ctx.cfg
        bank = rt.bank()
        if torch.is_grad_enabled():
            # create_graph=True (gradient penalty): the data gradient must itself be differentiable; parameter gradients
            # of this first pass are not (the reference only ever takes it w.r.t. the input image, losses.py:268-275)
            if _param_grad_wanted(ctx.weight, ctx.bias):
                raise NotImplementedError("create_graph=True is supported for input gradients only (WGAN-GP path)")
            dx = ConvDgradFn.apply(dy, x, ctx.weight, rt, slot, cfg) if ctx.needs_input_grad[0] else None
            return dx, None, None, (dy if ctx.has_res else None), None, None, None, None
        dy = _c(dy)
        N, Hs, Ws, Cin = x.shape
        up = 2 if cfg.in_upsample else 1
        Hin, Win = Hs * up, Ws * up
        Ho = (Hin + 2 * cfg.pad_h - cfg.R) // cfg.stride + 1
        Wo = (Win + 2 * cfg.pad_w - cfg.S) // cfg.stride + 1
        pool = cfg.out_pool
        scale = 0.25 if pool else 1.0
        dx = None
        skip_dx = ctx.link.take() if ctx.link is not None else None     # the skip path's gradient w.r.t. this same input (GradLink)
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad(dy, x, rt, slot, cfg, res=skip_dx)
        elif skip_dx is not None:
            raise RuntimeError("GradLink: a skip gradient was handed over but this convolution's input needs no gradient")
        if ctx.link is not None and ctx.link.chain:
            ctx.link.pending -= 1
            if ctx.link.pending > 0 and dx is not None:      # not the last reader of x: the next one adds this in its own launch
                ctx.link.dx, dx = dx, None
        want_db = ctx.bias is not None and ctx.needs_input_grad[2]
        db_done = False
        if ctx.needs_input_grad[1]:
            # the halo weight-gradient kernel holds the dy fragments anyway: the bias gradient rides along (no separate pass over dy)
            g = ensure_grad(ctx.bias) if (want_db and rt.rows_pad == rt.rows) else None
            db_done = _conv_wgrad(x, dy, rt, slot, cfg, dbias=g)
        if want_db and not db_done:
            g = ensure_grad(ctx.bias)
            rows = dy.shape[0] * dy.shape[1] * dy.shape[2]
            L.call("sg_colsum", L.dt(dy), L.ptr(dy), dy.shape[3], None, 0, rows, rt.rows, L.ptr(g), 1.0, L.stream())
        dres = dy if ctx.has_res else None
        return dx, None, None, dres, None, None, None, None


class ConvSkipFn(torch.autograd.Function):
    """y = [avgpool2]( conv3x3(relu?(h)) + b2 + conv1x1(up2?(relu?(x))) + b0 ): the tail of a residual block -- its last 3x3 convolution and its
    1x1 skip convolution -- as ONE fused launch (conv_v4.h SKIP) when the kernel takes the shape, else as the two launches of ConvFn chained
    through the residual input. Reference: src/models/big_resnet.py:28-42 (GenBlock: skip on the nearest-upsampled block input),
    :221-242 (DiscBlock: main and skip both average-pooled; nn.ReLU(inplace=True) makes the skip see relu(x), see backbones/big_resnet.py).
    backward: the two data gradients and the two weight gradients of the unfused form (the fusion is forward-only)."""

    @staticmethod
    def forward(ctx, h, x, w2, b2, w0, b0, rt2, rt0, slot, cfg2, cfg0, link=None):
        bank = rt2.bank()
        ctx.link = link
        h, x = _c(h), _c(x)
        _tick()
        # (cfg0.in_relu may differ from cfg2.in_relu: the first discriminator block's skip reads the image itself, big_resnet.py:177-192)
        assert cfg2.R == 3 and cfg0.R == 1 and cfg2.out_pool == cfg0.out_pool and not cfg2.in_upsample and (cfg0.in_relu == cfg2.in_relu or not cfg0.in_relu)
        pf = L.PIX_RELU if cfg2.in_relu else 0
        same_relu = cfg0.in_relu == cfg2.in_relu
        ef = L.EPI_POOL if cfg2.out_pool else 0
        al = 0.25 if cfg2.out_pool else 1.0
        y = None
        plain = rt2.rows_pad == rt2.rows and rt0.rows_pad == rt0.rows and b2 is not None and b0 is not None
        # measured (tools/skip_bench.py, profiles/r03_skip_bench_c.txt): the fused launch wins from 16 x 16 outputs up (0.01-0.30 ms per block
        # tail at batch 256) and loses 0.04-0.06 ms on the 1536-channel 8 x 8 tails, whose 48 one-tap slices are all stop-and-go
        # (a pooled tail goes through the quad kernel -- 2.25 x fewer MFMAs than the fused 3x3 launch -- and the 1x1 skip adds itself as a residual launch)
        if plain and same_relu and h.dtype == torch.bfloat16 and _SKIP_FUSION[0] and (h.shape[1] >= 16 or _SKIP_FUSION[0] == "all") and _quad_form(rt2, cfg2, h) is None:
            # (None: the kernel does not take the shape -- sg_conv2d_fwd_skip_ok includes the launcher's LDS limit -- and the two-launch form below runs;
            # a launch that fails after that is a real fault and propagates)
            y = conv2d_skip_raw(h, bank.w_fwd(slot, rt2), h.shape[3], rt2.rows, x, bank.w_fwd(slot, rt0), x.shape[3], cfg0.in_upsample, pf, ef,
                                bias=b2, bias2=b0, alpha=al, stats=cfg2.stats and _BN_FUSED_STATS[0])
        if y is None and plain and _quad_form(rt2, cfg2, h) == L.Q_POOL and _SKIP_FUSION[0] and (rt0.cin_pad % 32 == 0 or rt0.cin_pad == 8) and not cfg0.in_upsample:
            # pooled tail on the quad kernel with the skip as extra one-tap K-slices of the same launch (conv_q.h SKIP; an 8-channel skip input --
            # the image -- is ONE slice holding its four parity views, filter image mode 5)
            y = conv2d_q_raw(h, bank.w_quad(slot, rt2, L.Q_POOL), L.Q_POOL, h.shape[3], rt2.rows, pf, 0, bias=b2,
                             x2=x, w2q_ptr=bank.w_quad(slot, rt0, 5 if rt0.cin_pad == 8 else 4), bias2=b0, x2_norelu=not same_relu)
        if y is None:
            hh = _conv_fwd(h, rt2, slot, cfg2, b2)
            pf0 = (L.PIX_RELU if cfg0.in_relu else 0) | (L.PIX_UPSAMPLE if cfg0.in_upsample else 0)
            y = conv2d_raw(x, bank.w_fwd(slot, rt0), x.shape[3], rt0.rows_pad, 1, 1, 1, 0, 0, pf0, ef, bias=b0, res=hh, alpha=al)
        ctx.save_for_backward(h, x)
        ctx.rt2, ctx.rt0, ctx.slot, ctx.cfg2, ctx.cfg0 = rt2, rt0, slot, cfg2, cfg0
        ctx.w2, ctx.b2, ctx.w0, ctx.b0 = w2, b2, w0, b0
        return y

    @staticmethod
    def backward(ctx, dy):
        h, x = ctx.saved_tensors
        rt2, rt0, slot, cfg2, cfg0 = ctx.rt2, ctx.rt0, ctx.slot, ctx.cfg2, ctx.cfg0
        if torch.is_grad_enabled():      # create_graph=True: differentiable data gradients only (see ConvFn.backward)
            if _param_grad_wanted(ctx.w2, ctx.b2, ctx.w0, ctx.b0):
                raise NotImplementedError("create_graph=True is supported for input gradients only (WGAN-GP path)")
            dh = ConvDgradFn.apply(dy, h, ctx.w2, rt2, slot, cfg2) if ctx.needs_input_grad[0] else None
            dx = ConvDgradFn.apply(dy, x, ctx.w0, rt0, slot, cfg0) if ctx.needs_input_grad[1] else None
            return (dh, dx) + (None,) * 10
        dy = _c(dy)
        bank = rt2.bank()
        outs = []
        # both biases see the same gradient (the column sums of dy): when the halo weight-gradient kernel of the 3x3 convolution produces it
        # on the side, it goes to a scratch vector that is then added to BOTH bias gradients -- the 1x1's own pass over dy (sg_colsum) is gone
        want2 = ctx.b2 is not None and ctx.needs_input_grad[3]
        want0 = ctx.b0 is not None and ctx.needs_input_grad[5]
        shared_db = None
        if want2 and want0 and ctx.needs_input_grad[2] and ctx.needs_input_grad[4] and rt2.rows_pad == rt2.rows and rt0.rows == rt2.rows:
            shared_db = torch.zeros(rt2.rows, dtype=torch.float32, device=dy.device)
        for inp, rt, cfg, w_i, b_i, wp, bp in ((h, rt2, cfg2, 2, 3, ctx.w2, ctx.b2), (x, rt0, cfg0, 4, 5, ctx.w0, ctx.b0)):
            k = 0 if inp is h else 1
            N, Hs, Ws, Cin = inp.shape
            up = 2 if cfg.in_upsample else 1
            Ho, Wo = Hs * up, Ws * up      # 3x3 pad 1 / 1x1 pad 0, stride 1
            pool = cfg.out_pool
            outs.append(_conv_dgrad(dy, inp, rt, slot, cfg) if ctx.needs_input_grad[k] else None)
            want_db = bp is not None and ctx.needs_input_grad[b_i]
            db_done = False
            if ctx.needs_input_grad[w_i]:
                xf = (L.PIX_RELU if cfg.in_relu else 0) | (L.PIX_UPSAMPLE if cfg.in_upsample else 0)
                g = ensure_grad(bp) if (want_db and rt.rows_pad == rt.rows) else None
                if k == 0 and shared_db is not None:
                    g = shared_db
                elif k == 1 and shared_db is not None:
                    g = None                     # filled from the shared vector below
                db_done = _conv_wgrad(inp, dy, rt, slot, cfg, dbias=g)
                if k == 0 and shared_db is not None:
                    if db_done:
                        for bq in (ctx.b2, ctx.b0):
                            L.call("sg_axpby", L.F32, L.ptr(shared_db), L.ptr(ensure_grad(bq)), rt2.rows, 1.0, 1.0, L.stream())
                        continue
                    shared_db = None             # the kernel did not fuse the bias gradient: each convolution runs its own column sum
                elif k == 1 and shared_db is not None:
                    continue
            if want_db and not db_done:
                g = ensure_grad(bp)
                L.call("sg_colsum", L.dt(dy), L.ptr(dy), dy.shape[3], None, 0, dy.shape[0] * dy.shape[1] * dy.shape[2], rt.rows, L.ptr(g), 1.0, L.stream())
        if ctx.link is not None and outs[1] is not None and _GRAD_LINK[0]:
            # the block's first operator consumes x as well and runs its backward after this one: it adds this gradient in its own launch
            ctx.link.dx = outs[1]
            outs[1] = None
        return (outs[0], outs[1]) + (None,) * 10


_SKIP_FUSION = [{"0": False, "all": "all"}.get(os.environ.get("SG_SKIP_FUSION", "1"), True)]      # tests / A-B runs: SG_SKIP_FUSION=0 (or functional._SKIP_FUSION[0] = False) forces the two-launch form


class SliceUpFn(torch.autograd.Function):
    """y = nearest_up(x[..., :C]) (up in {1, 2}): the channel-slice skip of a BigGAN-deep generator block
    (reference src/models/big_resnet_deep_legacy.py:53-56,74-75). link: a GradLink shared with the block's first operator (its bn1 reads the same x): the
    skip's gradient is stashed there and added inside that operator's backward launch instead of by a separate elementwise add."""

    @staticmethod
    def forward(ctx, x, C, up, link=None):
        x = _c(x)
        N, Hs, Ws, ld = x.shape
        ctx.dims = (N, Hs, Ws, ld, C, up)
        ctx.link = link
        if C == ld and up == 1:          # the whole tensor at its own resolution: the identity skip of a non-resampling block (no launch)
            return x.view_as(x)
        y = torch.empty((N, Hs * up, Ws * up, C), dtype=x.dtype, device=x.device)
        L.call("sg_slice_up_fwd", L.dt(x), L.ptr(x), L.ptr(y), N, Hs, Ws, ld, C, up, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("SliceUpFn")
        N, Hs, Ws, ld, C, up = ctx.dims
        dy = _c(dy)
        if C == ld and up == 1:
            dx = dy
        else:
            dx = torch.empty((N, Hs, Ws, ld), dtype=dy.dtype, device=dy.device)
            L.call("sg_slice_up_bwd", L.dt(dy), L.ptr(dy), L.ptr(dx), N, Hs, Ws, ld, C, up, L.stream())
        if ctx.link is not None and _GRAD_LINK[0]:
            ctx.link.dx, dx = dx, None
        return dx, None, None, None


class CatConvFn(torch.autograd.Function):
    """out = cat([x, conv1x1(x) + bias], channel): the learnable channel-concat skip of a BigGAN-deep discriminator block
    (reference src/models/big_resnet_deep_legacy.py:236-238). The convolution writes straight into its channel slice; the
    backward reads the gradient slices in place (data gradient = one launch with the copied slice as its residual)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rt, slot):
        bank = rt.bank()
        x = _c(x)
        N, H, W, Cin = x.shape
        Cc = rt.rows
        out = torch.empty((N, H, W, Cin + Cc), dtype=x.dtype, device=x.device)
        L.call("sg_copy_channels", L.dt(x), L.ptr(x), Cin, L.ptr(out), Cin + Cc, N * H * W, Cin, L.stream())
        conv2d_raw(x, bank.w_fwd(slot, rt), Cin, Cc, 1, 1, bias=bias, out=out, out_coff=Cin)
        ctx.save_for_backward(x)
        ctx.rt, ctx.slot, ctx.bias = rt, slot, bias
        ctx.weight = weight      # the master parameter: only handed on to CatConvDgradFn so the second-order graph reaches it
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        rt, slot = ctx.rt, ctx.slot
        if torch.is_grad_enabled():
            # create_graph=True (R1 / gradient penalties on a BigGAN-deep discriminator): the data gradient as a differentiable operator
            if _param_grad_wanted(ctx.weight, ctx.bias):
                raise NotImplementedError("create_graph=True is supported for input gradients only (WGAN-GP / R1 path)")
            return (CatConvDgradFn.apply(dy, ctx.weight, rt, slot, x.shape[3]) if ctx.needs_input_grad[0] else None), None, None, None, None
        bank = rt.bank()
        dy = _c(dy)
        N, H, W, Cin = x.shape
        Cc = rt.rows
        ld = Cin + Cc
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_raw(dy, bank.w_dgrad(slot, rt), Cc, Cin, 1, 1, res=dy, ldx=ld, x_coff=Cin)
        if ctx.needs_input_grad[1]:
            conv2d_wgrad_raw(x, dy, bank.dwt(slot, rt), Cin, Cc, 1, 1, H, W, ldg=ld, dy_coff=Cin)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            g = ensure_grad(ctx.bias)
            L.call("sg_colsum", L.dt(dy), L.ptr(dy) + Cin * dy.element_size(), ld, None, 0, N * H * W, Cc, L.ptr(g), 1.0, L.stream())
        return dx, None, None, None, None


class CatConvDgradFn(torch.autograd.Function):
    """dx = dy[..., :Cin] + W^T dy[..., Cin:]: CatConvFn's data gradient as a differentiable operator (second-order pass). Linear in dy and in W:
        d/d(dy) = cat([t, conv1x1(t; W)])   -- CatConvFn's forward launches again, without the bias
        d/dW    = wgrad(t, dy[..., Cin:])    -- the forward's weight-gradient launch with x := t"""

    @staticmethod
    def forward(ctx, dy, weight, rt, slot, Cin):
        dy = _c(dy)
        ctx.save_for_backward(dy)
        ctx.rt, ctx.slot, ctx.Cin = rt, slot, Cin
        Cc = rt.rows
        return conv2d_raw(dy, rt.bank().w_dgrad(slot, rt), Cc, Cin, 1, 1, res=dy, ldx=Cin + Cc, x_coff=Cin)

    @staticmethod
    def backward(ctx, ddx):
        (dy,) = ctx.saved_tensors
        rt, slot, Cin = ctx.rt, ctx.slot, ctx.Cin
        bank = rt.bank()
        t = _c(ddx)
        N, H, W, _ = t.shape
        Cc = rt.rows
        g_dy = None
        if ctx.needs_input_grad[0]:
            g_dy = torch.empty((N, H, W, Cin + Cc), dtype=t.dtype, device=t.device)
            L.call("sg_copy_channels", L.dt(t), L.ptr(t), Cin, L.ptr(g_dy), Cin + Cc, N * H * W, Cin, L.stream())
            conv2d_raw(t, bank.w_fwd(slot, rt), Cin, Cc, 1, 1, out=g_dy, out_coff=Cin)
        if ctx.needs_input_grad[1]:
            conv2d_wgrad_raw(t, dy, bank.dwt(slot, rt), Cin, Cc, 1, 1, H, W, ldg=Cin + Cc, dy_coff=Cin)
        return g_dy, None, None, None, None


class ConvTransposeFn(torch.autograd.Function):
    """nn.ConvTranspose2d (reference src/utils/ops.py:176-184,207-216; DCGAN generator, src/models/deep_conv.py:21).
    forward = transposed gather on the engine; data gradient = the ordinary strided convolution of dy; weight gradient =
    the convolution weight-gradient kernel with the roles of x and dy exchanged (result lands as [Cin][R][S][Cout])."""

    @staticmethod
    def forward(ctx, x, weight, bias, rt, slot, cfg):
        bank = rt.bank()
        x = _c(x)
        N, H, W, Cin = x.shape
        assert Cin == rt.Cin
        Ho = (H - 1) * cfg.stride - 2 * cfg.pad_h + cfg.R
        Wo = (W - 1) * cfg.stride - 2 * cfg.pad_w + cfg.S
        y = conv2d_raw(x, bank.w_fwd(slot, rt), Cin, rt.rows, cfg.R, cfg.S, cfg.stride, cfg.pad_h, cfg.pad_w, L.PIX_TRANSPOSED, 0, bias=bias,
                       transposed_out_hw=(Ho, Wo))
        ctx.save_for_backward(x)
        ctx.rt, ctx.slot, ctx.cfg, ctx.bias = rt, slot, cfg, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("ConvTransposeFn")
        (x,) = ctx.saved_tensors
        rt, slot, cfg = ctx.rt, ctx.slot, ctx.cfg
        bank = rt.bank()
        dy = _c(dy)
        N, H, W, Cin = x.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_raw(dy, bank.w_dgrad(slot, rt), rt.rows, Cin, cfg.R, cfg.S, cfg.stride, cfg.pad_h, cfg.pad_w)
        if ctx.needs_input_grad[1]:
            conv2d_wgrad_raw(dy, x, bank.dwt(slot, rt), rt.rows, Cin, cfg.R, cfg.S, H, W, cfg.stride, cfg.pad_h, cfg.pad_w)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            g = ensure_grad(ctx.bias)
            L.call("sg_colsum", L.dt(dy), L.ptr(dy), dy.shape[3], None, 0, dy.shape[0] * dy.shape[1] * dy.shape[2], rt.rows, L.ptr(g), 1.0, L.stream())
        return dx, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """y = x W_sn^T + b in fp32 (nn.Linear, reference src/utils/ops.py:187-188,219-220). const_bias: non-trainable bias
    vector (the '1 +' of ConditionalBatchNorm2d's gain, reference src/utils/ops.py:25)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rt, slot, const_bias):
        bank = rt.bank()
        x = _c(x.float())
        B, K = x.shape
        assert K == rt.cols
        y = torch.empty((B, rt.rows), dtype=torch.float32, device=x.device)
        b = bias if bias is not None else const_bias
        gemm_raw(L.F32, bank.w_f32(slot, rt), 0, K, x, 0, K, y, rt.rows, rt.rows, B, K, bias=b)
        ctx.save_for_backward(x)
        ctx.rt, ctx.slot, ctx.bias = rt, slot, bias
        ctx.weight = weight      # the master parameter: only handed on to LinearDgradFn so the second-order graph reaches it
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        rt, slot = ctx.rt, ctx.slot
        if torch.is_grad_enabled():      # create_graph=True: the data gradient as a differentiable operator; parameter gradients of this first pass are not wanted
            if _param_grad_wanted(ctx.weight, ctx.bias):
                raise NotImplementedError("create_graph=True is supported for input gradients only (gradient penalties, latent optimisation)")
            return (LinearDgradFn.apply(dy, ctx.weight, rt, slot) if ctx.needs_input_grad[0] else None), None, None, None, None, None
        bank = rt.bank()
        dy = _c(dy.float())
        B, K = x.shape
        O = rt.rows
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, K), dtype=torch.float32, device=x.device)
            # dx[b][k] = sum_o dy[b][o] W[o][k] : P(i=k, red=o) = W stored [o][k] -> row-contiguous form
            gemm_dgrad_rows(bank.w_f32(slot, rt), dy, dx, B, K, O)
        if ctx.needs_input_grad[1]:
            # dW[o][k] += sum_b dy[b][o] x[b][k]  (accumulated into the slot's zero-initialised scratch: in a pass that follows a create_graph pass through
            # the same forward -- latent optimisation -- LinearDgradFn.backward has already put its share there)
            dw = bank.dwt(slot, rt)
            gemm_raw(L.F32, x, 1, K, dy, 1, O, dw, K, K, O, B, res=dw, ldr=K)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            g = ensure_grad(ctx.bias)
            L.call("sg_colsum", L.F32, L.ptr(dy), O, None, 0, B, O, L.ptr(g), 1.0, L.stream())
        return dx, None, None, None, None, None


class LinearDgradFn(torch.autograd.Function):
    """dx = dy W_sn: LinearFn's data gradient as a differentiable operator (second-order pass through a generator: latent optimisation, reference
    src/utils/losses.py:278-298). Linear in dy and in W: d/d(dy) = t W_sn^T (the forward without bias), d/dW = dy^T t into the bank's scratch."""

    @staticmethod
    def forward(ctx, dy, weight, rt, slot):
        dy = _c(dy.float())
        B, O = dy.shape
        assert O == rt.rows
        ctx.save_for_backward(dy)
        ctx.rt, ctx.slot = rt, slot
        dx = torch.empty((B, rt.cols), dtype=torch.float32, device=dy.device)
        gemm_dgrad_rows(rt.bank().w_f32(slot, rt), dy, dx, B, rt.cols, O)
        return dx

    @staticmethod
    def backward(ctx, ddx):
        (dy,) = ctx.saved_tensors
        rt, slot = ctx.rt, ctx.slot
        bank = rt.bank()
        t = _c(ddx.float())
        B, K = t.shape
        O = rt.rows
        g_dy = None
        if ctx.needs_input_grad[0]:
            g_dy = torch.empty((B, O), dtype=torch.float32, device=t.device)
            gemm_raw(L.F32, bank.w_f32(slot, rt), 0, K, t, 0, K, g_dy, O, O, B, K)
        if ctx.needs_input_grad[1]:
            dw = bank.dwt(slot, rt)
            gemm_raw(L.F32, t, 1, K, dy, 1, O, dw, K, K, O, B, res=dw, ldr=K)      # dW[o][k] += sum_b dy[b][o] t[b][k]
        return g_dy, None, None, None


class CbnAffineFn(torch.autograd.Function):
    """[1 + gain(y) | bias(y)] of a ConditionalBatchNorm2d (reference src/utils/ops.py:21-27: two (sn)linear layers without bias on the same
    conditioning vector) as ONE fp32 GEMM over the 2 C rows of the two weight images, which sit back to back in the network's bank: the two
    linears were ~125 launches of 28-42 us per C3 step (forward, data gradient, weight gradient). Returns the packed [B][2 C] tensor BNFn takes
    with cfg.packed; falls back to two GEMMs writing the two halves when the images are not adjacent."""

    @staticmethod
    def forward(ctx, y, wg, wb, rt_g, rt_b, slot, const2):
        bank = rt_g.bank()
        y = _c(y.float())
        B, K = y.shape
        C = rt_g.rows
        assert K == rt_g.cols == rt_b.cols and rt_b.rows == C
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=y.device)
        pg, pb = bank.w_f32(slot, rt_g), bank.w_f32(slot, rt_b)
        ctx.adjacent = pb == pg + 4 * C * K
        if ctx.adjacent:
            gemm_raw(L.F32, pg, 0, K, y, 0, K, out, 2 * C, 2 * C, B, K, bias=const2)
        else:
            gemm_raw(L.F32, pg, 0, K, y, 0, K, out, 2 * C, C, B, K, bias=const2)
            gemm_raw(L.F32, pb, 0, K, y, 0, K, out.data_ptr() + 4 * C, 2 * C, C, B, K)
        ctx.save_for_backward(y)
        ctx.rt_g, ctx.rt_b, ctx.slot = rt_g, rt_b, slot
        return out

    @staticmethod
    def backward(ctx, dgb):
        _first_order_only("CbnAffineFn")
        (y,) = ctx.saved_tensors
        rt_g, rt_b, slot = ctx.rt_g, ctx.rt_b, ctx.slot
        bank = rt_g.bank()
        dgb = _c(dgb.float())
        B, K = y.shape
        C = rt_g.rows
        pg, pb = bank.w_f32(slot, rt_g), bank.w_f32(slot, rt_b)
        # (a frozen half -- only one of the two weights requires a gradient -- gets none: the merged 2 C-row GEMM is taken when BOTH want theirs)
        dg = bank.dwt(slot, rt_g) if ctx.needs_input_grad[1] else None
        db = bank.dwt(slot, rt_b) if ctx.needs_input_grad[2] else None
        both = dg is not None and db is not None
        adjacent = ctx.adjacent and (not both or db == dg + 4 * C * K)
        dy = None
        if ctx.needs_input_grad[0]:
            dy = torch.empty((B, K), dtype=torch.float32, device=y.device)
            if adjacent:      # dy[b][k] = sum over the 2 C rows of dgb[b][o] W[o][k]
                gemm_dgrad_rows(pg, dgb, dy, B, K, 2 * C)
            else:
                gemm_raw(L.F32, pg, 1, K, dgb, 0, 2 * C, dy, K, K, B, C)
                gemm_raw(L.F32, pb, 1, K, dgb.data_ptr() + 4 * C, 0, 2 * C, dy, K, K, B, C, res=dy, ldr=K)
        if both and adjacent:      # dW[o][k] = sum_b dgb[b][o] y[b][k], o over the 2 C rows
            gemm_raw(L.F32, y, 1, K, dgb, 1, 2 * C, dg, K, K, 2 * C, B)
        else:
            if dg is not None:
                gemm_raw(L.F32, y, 1, K, dgb, 1, 2 * C, dg, K, K, C, B)
            if db is not None:
                gemm_raw(L.F32, y, 1, K, dgb.data_ptr() + 4 * C, 1, 2 * C, db, K, K, C, B)
        return dy, None, None, None, None, None, None


class EmbeddingFn(torch.autograd.Function):
    """Plain (non-SN) embedding lookup, e.g. G's shared class embedding (reference src/models/big_resnet.py:98,136)."""

    @staticmethod
    def forward(ctx, weight, idx):
        idx = _c(idx.long())
        B = idx.numel()
        num, dim = weight.shape
        out = torch.empty((B, dim), dtype=torch.float32, device=weight.device)
        L.call("sg_embedding_fwd", L.ptr(weight), L.ptr(idx), L.ptr(out), B, dim, num, L.stream())
        ctx.save_for_backward(idx)
        ctx.weight = weight
        return out

    @staticmethod
    def backward(ctx, dout):
        _first_order_only("EmbeddingFn")
        (idx,) = ctx.saved_tensors
        w = ctx.weight
        if ctx.needs_input_grad[0]:
            g = ensure_grad(w)
            dout = _c(dout.float())
            L.call("sg_embedding_bwd", L.ptr(dout), L.ptr(idx), L.ptr(g), idx.numel(), w.shape[1], w.shape[0], L.stream())
        return None, None


class SNEmbeddingFn(torch.autograd.Function):
    """Embedding lookup in the spectrally normalised table held by the bank (sn_embedding, reference ops.py:223-224)."""

    @staticmethod
    def forward(ctx, weight, idx, rt, slot):
        bank = rt.bank()
        idx = _c(idx.long())
        B = idx.numel()
        out = torch.empty((B, rt.cols), dtype=torch.float32, device=weight.device)
        L.call("sg_embedding_fwd", bank.w_f32(slot, rt), L.ptr(idx), L.ptr(out), B, rt.cols, rt.rows, L.stream())
        ctx.save_for_backward(idx)
        ctx.rt, ctx.slot = rt, slot
        return out

    @staticmethod
    def backward(ctx, dout):
        _first_order_only("SNEmbeddingFn")
        (idx,) = ctx.saved_tensors
        rt, slot = ctx.rt, ctx.slot
        if ctx.needs_input_grad[0]:
            dout = _c(dout.float())
            L.call("sg_embedding_bwd", L.ptr(dout), L.ptr(idx), rt.bank().dwt(slot, rt), idx.numel(), rt.cols, rt.rows, L.stream())
        return None, None, None, None


# ---------------------------------------------------------------------------------------------------------
# (conditional) batch norm
# ---------------------------------------------------------------------------------------------------------
class BNCfg:
    __slots__ = ("batch_stats", "track", "eps", "momentum", "relu", "group", "packed")

    def __init__(self, batch_stats, track, eps, momentum, relu, group=None, packed=False):
        self.batch_stats, self.track, self.eps, self.momentum, self.relu, self.group = batch_stats, track, eps, momentum, relu, group
        self.packed = packed      # `gain` is the packed per-sample [N][gain(C) | bias(C)] tensor of a conditional batch norm (CbnAffineFn), `bias` is None


def _world(group):
    if group is not None and dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group if group is not True else None)
    return 1


def _allreduce_sum(t, group):
    """sum over the data-parallel ranks: the C ABI's RCCL entry point when a native communicator serves the group (comm.enable),
    torch.distributed otherwise."""
    nc = _comm.native_for(group)
    pp = _comm.p2p_for(group) if (t.dtype == torch.float64 and t.numel() <= _comm.P2PMailbox.MAX_DOUBLES) else None
    with _comm.exposed():          # on the compute stream: the whole collective is exposed (bench.py exposed_comm_ms_per_step)
        if pp is not None:         # peer-store mailboxes: one launch, one xGMI round trip (csrc/p2p.hip)
            pp.allreduce_f64_(t)
        elif nc is not None:
            nc.allreduce_(t)
        else:
            dist.all_reduce(t, group=None if group is True else group)


class BNFn(torch.autograd.Function):
    """y = relu?( (x - mean) * invstd * gain + bias ) with batch or running statistics.

    gain/bias: None, per-channel [C] (nn.BatchNorm2d affine) or per-sample [N,C] (ConditionalBatchNorm2d: gain already
    holds 1 + W_g y). Sync-BN = one all-reduce of the fp64 partial sums between the two kernels
    (reference src/utils/ops.py:14-28,227-228; src/models/model.py:161-165)."""

    @staticmethod
    def forward(ctx, x, gain, bias, running_mean, running_var, cfg, *opt):
        x = _c(x)
        _tick()
        fused = _take_stats(x) if cfg.batch_stats else None      # statistics the producing convolution took in its epilogue
        ctx.link = opt[0] if opt else None      # optional 7th argument: a GradLink (see ConvSkipFn)
        ctx.nopt = len(opt)
        N, H, W, Cc = x.shape
        HW = H * W
        dev = x.device
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
        count = float(N * HW)
        if cfg.batch_stats:
            ws = _world(cfg.group)
            rm = running_mean if cfg.track else None
            rv = running_var if cfg.track else None
            nc = _comm.native_for(cfg.group) if ws > 1 else None
            pp = _comm.p2p_for(cfg.group) if (ws > 1 and 2 * Cc <= _comm.P2PMailbox.MAX_DOUBLES) else None
            if pp is not None:
                # sync-BN with the exchange FUSED INTO the finalize kernel: this rank's partial sums go straight into every peer's HBM (csrc/p2p.hip), the same
                # launch waits for the peers' and writes mean / invstd / running statistics of the global batch
                partial = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
                if fused is not None:
                    L.call("sg_bn_stats_from_tiles", fused[0].data_ptr(), fused[1], Cc, L.ptr(partial), L.stream())
                else:
                    L.call("sg_bn_partial_stats", L.dt(x), L.ptr(x), Cc, N * HW, Cc, L.ptr(partial), L.stream())
                count *= ws
                with _comm.exposed():
                    L.call("sg_bn_finalize_p2p", pp.handle, L.ptr(partial), count, Cc, cfg.eps, cfg.momentum, L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.stream())
            elif fused is not None:
                partial = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
                L.call("sg_bn_stats_from_tiles", fused[0].data_ptr(), fused[1], Cc, L.ptr(partial), L.stream())
                if ws > 1:
                    _allreduce_sum(partial, cfg.group)
                    count *= ws
                L.call("sg_bn_finalize", L.ptr(partial), count, Cc, cfg.eps, cfg.momentum, L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.stream())
            elif nc is not None:
                # sync-BN statistics in ONE C-ABI call: partial sums -> RCCL all-reduce -> mean / invstd / running stats, same stream
                partial = torch.empty(2 * Cc, dtype=torch.float64, device=dev)
                L.call("sg_bn_stats_sync", L.dt(x), L.ptr(x), Cc, N * HW, Cc, L.ptr(partial), nc.handle, cfg.eps, cfg.momentum, L.ptr(mean),
                       L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.stream())
                count *= ws
            else:
                partial = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
                L.call("sg_bn_partial_stats", L.dt(x), L.ptr(x), Cc, N * HW, Cc, L.ptr(partial), L.stream())
                if ws > 1:
                    dist.all_reduce(partial, group=None if cfg.group is True else cfg.group)
                    count *= ws
                L.call("sg_bn_finalize", L.ptr(partial), count, Cc, cfg.eps, cfg.momentum, L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.stream())
        else:
            L.call("sg_bn_from_running", L.ptr(running_mean), L.ptr(running_var), Cc, cfg.eps, L.ptr(mean), L.ptr(invstd), L.stream())
        gsn = 0
        bias_ptr_off = 0
        if cfg.packed:
            gain = _c(gain.float())
            assert bias is None and gain.dim() == 2 and gain.shape[1] == 2 * Cc
            gsn, bias_ptr_off = 2 * Cc, 4 * Cc
        elif gain is not None:
            gain = _c(gain.float())
            gsn = Cc if gain.dim() == 2 else 0
        if bias is not None:
            bias = _c(bias.float())
            assert (Cc if bias.dim() == 2 else 0) == gsn or gain is None
            if gain is None:
                gsn = Cc if bias.dim() == 2 else 0
        y = torch.empty_like(x)
        bptr = (L.ptr(gain) + bias_ptr_off) if cfg.packed else L.ptr(bias)
        L.call("sg_bn_apply", L.dt(x), L.ptr(x), L.ptr(y), N, HW, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), bptr, gsn, 1 if cfg.relu else 0, L.stream())
        ctx.save_for_backward(x, gain, bias, mean, invstd)
        ctx.cfg, ctx.gsn, ctx.count = cfg, gsn, count
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gain, bias, mean, invstd = ctx.saved_tensors
        cfg, gsn = ctx.cfg, ctx.gsn
        if torch.is_grad_enabled():
            if gsn or _param_grad_wanted(gain, bias):
                raise NotImplementedError("create_graph=True is supported for the input gradient of per-channel BN only (WGAN-GP path)")
            dx = BNBwdFn.apply(dy, x, gain, bias, mean, invstd, cfg, ctx.count) if ctx.needs_input_grad[0] else None
            return (dx, None, None, None, None, None) + (None,) * ctx.nopt
        dy = _c(dy)
        N, H, W, Cc = x.shape
        HW = H * W
        dev = x.device
        skip_dx = ctx.link.take() if ctx.link is not None else None     # the skip path's gradient w.r.t. this same input (GradLink)
        if skip_dx is not None and not ctx.needs_input_grad[0]:
            raise RuntimeError("GradLink: a skip gradient was handed over but this batch norm's input needs no gradient")
        sums = torch.zeros((N, Cc, 2), dtype=torch.float32, device=dev)
        bptr = (L.ptr(gain) + 4 * Cc) if cfg.packed else L.ptr(bias)       # packed cBN rows: [gain(C) | bias(C)], pitch gsn = 2 C
        L.call("sg_bn_bwd_reduce", L.dt(x), L.ptr(x), L.ptr(dy), N, HW, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), bptr, gsn,
               1 if cfg.relu else 0, L.ptr(sums), L.stream())
        chan = torch.empty(2 * Cc, dtype=torch.float64, device=dev)
        dgain = torch.zeros_like(gain) if (gain is not None and ctx.needs_input_grad[1]) else None
        dbias = torch.zeros_like(bias) if (bias is not None and ctx.needs_input_grad[2]) else None
        dbptr = ((L.ptr(dgain) + 4 * Cc) if dgain is not None else None) if cfg.packed else L.ptr(dbias)
        L.call("sg_bn_bwd_finalize", L.ptr(sums), N, Cc, L.ptr(gain), gsn, L.ptr(dgain), dbptr, L.ptr(chan), L.stream())
        dx = None
        if ctx.needs_input_grad[0]:
            if cfg.batch_stats and _world(cfg.group) > 1:
                _allreduce_sum(chan, cfg.group)
            dx = torch.empty_like(x)
            L.call("sg_bn_bwd_apply_res", L.dt(x), L.ptr(x), L.ptr(dy), L.ptr(dx), N, HW, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), bptr,
                   gsn, 1 if cfg.relu else 0, L.ptr(chan), ctx.count, 1 if cfg.batch_stats else 0, L.ptr(_c(skip_dx) if skip_dx is not None else None), L.stream())
        return (dx, dgain, dbias, None, None, None) + (None,) * ctx.nopt


class BNBwdFn(torch.autograd.Function):
    """BN's data gradient dx(dy, x, gain) as a differentiable op (statistics are functions of x): the second-order pass
    of the gradient penalty through a discriminator that uses batch norm (WGAN-GP.yaml: no SN => BN in D). Formulas and
    kernels: csrc/norm.hip "second-order backward"."""

    @staticmethod
    def forward(ctx, dy, x, gain, bias, mean, invstd, cfg, count):
        dy = _c(dy)
        N, H, W, Cc = x.shape
        dev = x.device
        sums = torch.zeros((N, Cc, 2), dtype=torch.float32, device=dev)
        L.call("sg_bn_bwd_reduce", L.dt(x), L.ptr(x), L.ptr(dy), N, H * W, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), L.ptr(bias), 0,
               1 if cfg.relu else 0, L.ptr(sums), L.stream())
        chan = torch.empty(2 * Cc, dtype=torch.float64, device=dev)
        L.call("sg_bn_bwd_finalize", L.ptr(sums), N, Cc, L.ptr(gain), 0, None, None, L.ptr(chan), L.stream())
        if cfg.batch_stats and _world(cfg.group) > 1:
            _allreduce_sum(chan, cfg.group)
        dx = torch.empty_like(x)
        L.call("sg_bn_bwd_apply", L.dt(x), L.ptr(x), L.ptr(dy), L.ptr(dx), N, H * W, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), L.ptr(bias),
               0, 1 if cfg.relu else 0, L.ptr(chan), count, 1 if cfg.batch_stats else 0, L.stream())
        ctx.save_for_backward(dy, x, gain, bias, mean, invstd)
        ctx.cfg, ctx.count = cfg, count
        return dx

    @staticmethod
    def backward(ctx, u):
        dy, x, gain, bias, mean, invstd = ctx.saved_tensors
        cfg = ctx.cfg
        u = _c(u)
        N, H, W, Cc = x.shape
        dev = x.device
        relu = 1 if cfg.relu else 0
        sums = torch.zeros((N, Cc, 5), dtype=torch.float32, device=dev)
        L.call("sg_bn_bwd2_reduce", L.dt(x), L.ptr(x), L.ptr(dy), L.ptr(u), N, H * W, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), L.ptr(bias), relu,
               L.ptr(sums), L.stream())
        chan_local = torch.empty(5 * Cc, dtype=torch.float64, device=dev)
        L.call("sg_bn_bwd2_finalize", L.ptr(sums), N, Cc, L.ptr(chan_local), L.stream())
        chan = chan_local
        if cfg.batch_stats and _world(cfg.group) > 1:
            chan = chan_local.clone()
            _allreduce_sum(chan, cfg.group)
        use_batch = 1 if cfg.batch_stats else 0
        g_dy = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g_x = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        if g_dy is not None or g_x is not None:
            L.call("sg_bn_bwd2_apply", L.dt(x), L.ptr(x), L.ptr(dy), L.ptr(u), L.ptr(g_dy), L.ptr(g_x), N, H * W, Cc, L.ptr(mean), L.ptr(invstd),
                   L.ptr(gain), L.ptr(bias), relu, L.ptr(chan), ctx.count, use_batch, L.stream())
        dgain = None
        if gain is not None and ctx.needs_input_grad[2]:
            dgain = torch.zeros_like(gain)
            L.call("sg_bn_bwd2_dgain", L.ptr(chan_local), L.ptr(chan), ctx.count, L.ptr(invstd), Cc, use_batch, L.ptr(dgain), L.stream())
        return g_dy, g_x, dgain, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------------
# small elementwise ops of the D blocks
# ---------------------------------------------------------------------------------------------------------
class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.call("sg_avgpool2_fwd", L.dt(x), L.ptr(x), L.ptr(y), N, H, W, Cc, L.stream())
        ctx.shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        if torch.is_grad_enabled():
            return AvgPool2BwdFn.apply(dy, ctx.shape)
        return _avgpool2_bwd(dy, ctx.shape)


def _avgpool2_bwd(dy, shape):
    dy = _c(dy)
    N, H, W, Cc = shape
    dx = torch.empty((N, H, W, Cc), dtype=dy.dtype, device=dy.device)
    L.call("sg_avgpool2_bwd", L.dt(dy), L.ptr(dy), L.ptr(dx), N, H, W, Cc, L.stream())
    return dx


class AvgPool2BwdFn(torch.autograd.Function):
    """0.25 * broadcast of the pooled gradient; its adjoint is the pooling itself (second-order pass)."""

    @staticmethod
    def forward(ctx, dy, shape):
        ctx.shape = shape
        return _avgpool2_bwd(dy, shape)

    @staticmethod
    def backward(ctx, ddx):
        ddx = _c(ddx)
        N, H, W, Cc = ctx.shape
        g = torch.empty((N, H // 2, W // 2, Cc), dtype=ddx.dtype, device=ddx.device)
        L.call("sg_avgpool2_fwd", L.dt(ddx), L.ptr(ddx), L.ptr(g), N, H, W, Cc, L.stream())
        return g, None


class AddFn(torch.autograd.Function):
    """out = a + x (identity skip of a BN DiscBlock)."""

    @staticmethod
    def forward(ctx, a, x):
        a, x = _c(a), _c(x)
        out = torch.empty_like(a)
        L.call("sg_convert", L.dt(a), L.dt(a), L.ptr(a), L.ptr(out), a.numel(), L.stream())
        L.call("sg_axpby", L.dt(a), L.ptr(x), L.ptr(out), a.numel(), 1.0, 1.0, L.stream())
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def _mask(dy, x):
    dy = _c(dy)
    dx = torch.empty_like(x)
    L.call("sg_relu_mask", L.dt(x), L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), L.stream())
    return dx


class MaskFn(torch.autograd.Function):
    """dy * (x > 0) as a differentiable op of dy (x's mask is piecewise constant): the ReLU backward inside a create_graph pass."""

    @staticmethod
    def forward(ctx, dy, x):
        ctx.save_for_backward(x)
        return _mask(dy, x)

    @staticmethod
    def backward(ctx, dd):
        (x,) = ctx.saved_tensors
        return _mask(dd, x), None


class ReluFn(torch.autograd.Function):
    """standalone ReLU (only where no neighbouring launch can absorb it): y = x * (x > 0). link: a GradLink shared with another reader of the same x whose
    backward runs later (the first convolution of a BigGAN-deep discriminator block): the masked gradient is stashed there and rides as the residual of that
    convolution's data-gradient launch."""

    @staticmethod
    def forward(ctx, x, link=None):
        x = _c(x)
        y = torch.empty_like(x)
        L.call("sg_relu_mask", L.dt(x), L.ptr(x), L.ptr(x), L.ptr(y), x.numel(), L.stream())
        ctx.save_for_backward(x)
        ctx.link = link
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return MaskFn.apply(dy, x), None
        dx = _mask(dy, x)
        if ctx.link is not None and _GRAD_LINK[0]:
            ctx.link.dx, dx = dx, None
        return dx, None


class AddReluFn(torch.autograd.Function):
    """out = a + relu(x): the identity-skip DiscBlock (its in-place ReLU also rewrites the skip tensor;
    reference src/models/big_resnet.py:221-242 with nn.ReLU(inplace=True), src/config.py:476)."""

    @staticmethod
    def forward(ctx, a, x):
        a, x = _c(a), _c(x)
        out = torch.empty_like(a)
        L.call("sg_add_relu", L.dt(a), L.ptr(a), L.ptr(x), L.ptr(out), a.numel(), L.stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[1]:
            dx = MaskFn.apply(dy, x) if torch.is_grad_enabled() else _mask(dy, x)
        return dy, dx


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
        dw1_dummy = None if train_w else torch.zeros(Cc, dtype=torch.float32, device=h.device)   # kept alive until after the launch
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
        scratch = torch.zeros(Cc + 1, dtype=torch.float32, device=dev)
        hz = torch.zeros((B, Cc), dtype=torch.float32, device=dev)
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
            dw1_dummy = None if ctx.needs_input_grad[1] else torch.zeros(Cc, dtype=torch.float32, device=dev)   # kept alive until after the launch
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


# ---------------------------------------------------------------------------------------------------------
# InfoGAN's Q heads (csrc/ext/losses.hip)
# ---------------------------------------------------------------------------------------------------------
class ExpFn(torch.autograd.Function):
    """y = exp(x) in fp32 (the continuous code's variance head, reference src/models/big_resnet.py:377)"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x.float())
        y = torch.empty_like(x)
        L.call("sg_exp_fwd", L.ptr(x), L.ptr(y), x.numel(), L.stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("ExpFn")
        (y,) = ctx.saved_tensors
        dy = _c(dy.float())
        dx = torch.empty_like(y)
        L.call("sg_exp_bwd", L.ptr(dy), L.ptr(y), L.ptr(dx), y.numel(), L.stream())
        return dx


class NormalNllFn(torch.autograd.Function):
    """reference src/utils/losses.py:369-375 normal_nll_loss(x, mu, var): value and the gradients w.r.t. mu and var from one launch (x: the sampled code)"""

    @staticmethod
    def forward(ctx, x, mu, var):
        x, mu, var = _c(x.detach().float()), _c(mu.float()), _c(var.float())
        if x.shape != mu.shape or mu.shape != var.shape or x.dim() != 2:
            raise RuntimeError("normal_nll_loss: three [B, K] tensors expected")
        loss = torch.empty(1, dtype=torch.float32, device=mu.device)
        dmu, dvar = torch.empty_like(mu), torch.empty_like(var)
        L.call("sg_normal_nll", L.ptr(x), L.ptr(mu), L.ptr(var), x.shape[0], x.shape[1], L.ptr(loss), L.ptr(dmu), L.ptr(dvar), L.stream())
        ctx.save_for_backward(dmu, dvar)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dmu, dvar = ctx.saved_tensors
        return None, dmu * g, dvar * g
