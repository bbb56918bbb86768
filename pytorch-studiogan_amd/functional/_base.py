"""shared state of the autograd layer (statistics hand-off between a convolution epilogue and the batch norm behind it, environment switches) and the raw launch helpers over the C ABI (also used directly by the kernel-level tests)."""
import os

import torch
import torch.distributed as dist

from .. import _lib as L
from .. import comm as _comm
from ..bank import ensure_grad


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


# Small zero-initialised scratch (batch-norm partial sums, per-sample reduction buffers, head scratch): ~110 torch.zeros() per BigGAN-128 step, each its own
# fill launch (4-5 us of a busy stream; 4400 of them per WGAN-GP step). They are carved out of ONE zeroed block per device instead -- a fresh 8 MiB block (one fill)
# whenever the current one is used up; a piece is handed out once and never again, and the block lives as long as any piece of it does.
_ZERO_POOL = {}
_ZERO_POOL_BYTES = 8 << 20
_ZERO_POOL_ON = [os.environ.get("SG_ZERO_POOL", "1") != "0"]


def zeros_small(shape, dtype, device):
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    if not _ZERO_POOL_ON[0] or nbytes == 0 or nbytes > (_ZERO_POOL_BYTES >> 1) or torch.device(device).type != "cuda":
        return torch.zeros(shape, dtype=dtype, device=device)
    key = torch.device(device)
    ent = _ZERO_POOL.get(key)
    step = (nbytes + 255) & ~255
    if ent is None or ent[1] + step > _ZERO_POOL_BYTES:
        ent = _ZERO_POOL[key] = [torch.zeros(_ZERO_POOL_BYTES, dtype=torch.uint8, device=device), 0]
    off = ent[1]
    ent[1] = off + step
    return ent[0][off:off + nbytes].view(dtype).view(shape)


def zeros_like_small(t):
    return zeros_small(tuple(t.shape), t.dtype, t.device)


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


__all__ = ['L', 'zeros_small', 'zeros_like_small', '_BN_FUSED_STATS', '_CBN_MERGED', '_DGRAD_SPLITK', '_SEQ', '_STATS_OFFER', '_c', '_comm', '_first_order_only', '_offer_stats', '_param_grad_wanted', '_take_stats', '_tick', 'conv2d_q_raw', 'conv2d_q_wgrad_raw', 'conv2d_raw', 'conv2d_skip_raw', 'conv2d_wgrad_raw', 'dist', 'ensure_grad', 'f32_mode', 'gemm_dgrad_rows', 'gemm_raw', 'os', 'quad_pack_raw', 'torch']
