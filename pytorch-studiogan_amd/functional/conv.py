"""convolution / transposed convolution / linear / embedding autograd functions over the weight bank (csrc/conv*.h, wgrad*.h, gemm_core.h; reference src/utils/ops.py:165-224)."""
import ctypes

from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

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
            bias_k = zeros_small(rt.rows_pad, torch.float32, x.device)
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
            shared_db = zeros_small(rt2.rows, torch.float32, dy.device)
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
        dy = _cbn_affine_backward(y, dgb, ctx.rt_g, ctx.rt_b, ctx.slot, ctx.adjacent, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dy, None, None, None, None, None, None


def _cbn_affine_backward(y, dgb, rt_g, rt_b, slot, fwd_adjacent, need_y, need_g, need_b):
    """backward of one conditional batch norm's [1 + gain(y) | bias(y)] product: -> dy ([B][K] or None); the weight gradients go into the bank's scratch"""
    bank = rt_g.bank()
    dgb = _c(dgb.float())
    B, K = y.shape
    C = rt_g.rows
    pg, pb = bank.w_f32(slot, rt_g), bank.w_f32(slot, rt_b)
    # (a frozen half -- only one of the two weights requires a gradient -- gets none: the merged 2 C-row GEMM is taken when BOTH want theirs)
    dg = bank.dwt(slot, rt_g) if need_g else None
    db = bank.dwt(slot, rt_b) if need_b else None
    both = dg is not None and db is not None
    adjacent = fwd_adjacent and (not both or db == dg + 4 * C * K)
    dy = None
    if need_y:
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
    return dy


class CbnAffineGroupFn(torch.autograd.Function):
    """The [1 + gain(y) | bias(y)] products of ALL conditional batch norms of a generator forward in one launch (csrc/linear_group.hip): every conditioning
    vector is known when the forward starts (reference src/models/big_resnet.py:139-163), and each product on its own is a 32 us launch of a few workgroups.
    apply(slot, metas, yidx, *tensors): metas[i] = (rt_gain, rt_bias, const2) of layer i, yidx[i] = which conditioning vector it reads,
    tensors = (y_0 .. y_{G-1}, gain_w_0, bias_w_0, gain_w_1, ...). Returns one packed [B][2 C_i] tensor per layer (what BNFn takes with cfg.packed).
    The backward runs layer by layer (CbnAffineFn's launches) once every layer's gradient has arrived."""

    @staticmethod
    def forward(ctx, slot, metas, yidx, *tensors):
        G = len(tensors) - 2 * len(metas)
        ys = [_c(t.float()) for t in tensors[:G]]
        B, K = ys[0].shape
        bank = metas[0][0].bank()
        dev = ys[0].device
        n = len(metas)
        total = sum(2 * m[0].rows for m in metas)
        out_all = torch.empty(B * total, dtype=torch.float32, device=dev)
        arr = (L.LinearItem * (2 * n))()
        outs, adj, off, k = [], [], 0, 0
        for i, (rt_g, rt_b, const2) in enumerate(metas):
            C = rt_g.rows
            y = ys[yidx[i]]
            assert y.shape == (B, K) and rt_g.cols == K and rt_b.cols == K and rt_b.rows == C
            pg, pb = bank.w_f32(slot, rt_g), bank.w_f32(slot, rt_b)
            a = pb == pg + 4 * C * K
            adj.append(a)
            base = out_all.data_ptr() + 4 * off
            parts = [(pg, 2 * C, const2.data_ptr(), base)] if a else [(pg, C, const2.data_ptr(), base), (pb, C, const2.data_ptr() + 4 * C, base + 4 * C)]
            for (w, rows, bias, o) in parts:
                it = arr[k]
                it.w, it.y, it.bias, it.out, it.rows, it.K, it.ldy, it.ldo = w, y.data_ptr(), bias, o, rows, K, K, 2 * C
                k += 1
            outs.append(out_all[off:off + B * 2 * C].view(B, 2 * C))
            off += B * 2 * C
        # the device copy of the table is kept per set of addresses: in steady state the caching allocator hands every step the same blocks, so the upload
        # happens during warm-up only (and from pinned memory: L.upload_bytes)
        raw = bytes(bytearray(arr)[:k * ctypes.sizeof(L.LinearItem)])
        cache = bank.__dict__.setdefault("_linear_group_tabs", {})
        tab = cache.get(raw)
        if tab is None:
            if len(cache) >= 64:
                cache.clear()
            tab = cache[raw] = L.upload_bytes(raw, dev)
        L.call("sg_linear_group", tab.data_ptr(), arr, k, B, L.stream())
        ctx.save_for_backward(*ys)
        ctx.metas, ctx.yidx, ctx.slot, ctx.adj, ctx.G = metas, yidx, slot, adj, G
        ctx._keep = out_all
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dgbs):
        _first_order_only("CbnAffineGroupFn")
        ys = ctx.saved_tensors
        G, n = ctx.G, len(ctx.metas)
        dys = [None] * G
        for i, ((rt_g, rt_b, _), dgb) in enumerate(zip(ctx.metas, dgbs)):
            if dgb is None:
                continue
            g = ctx.yidx[i]
            dy = _cbn_affine_backward(ys[g], dgb, rt_g, rt_b, ctx.slot, ctx.adj[i], ctx.needs_input_grad[3 + g],
                                      ctx.needs_input_grad[3 + G + 2 * i], ctx.needs_input_grad[3 + G + 2 * i + 1])
            if dy is not None:
                dys[g] = dy if dys[g] is None else dys[g].add_(dy)
        return (None, None, None) + tuple(dys) + (None,) * (2 * n)


def cbn_prefetch(slot, pairs):
    """pairs = [(ConditionalBatchNorm2d module, conditioning vector)] of a generator forward, in any order: one launch computes every layer's packed
    [1 + gain | bias] rows; each module's forward_nhwc then finds its rows in slot.cbn_rows (ops.ConditionalBatchNorm2d). Layers that do not qualify
    (a linear with a bias, SG_CBN_MERGED=0 / SG_CBN_GROUP=0) are left to their own launches."""
    slot.cbn_rows = {}
    if not (_CBN_GROUP[0] and _CBN_MERGED[0]):
        return
    ok = [(m, y) for (m, y) in pairs if m.gain.bias is None and m.bias.bias is None and m.gain.__dict__.get("_sg_rt") is not None]
    if len(ok) < 2:
        return
    if ok[0][0].gain._sg_rt.bank().exchange is not None:
        # early gradient exchange (optim.ExchangePlan, SG_EARLY_EXCHANGE=1): a block's arena range is sent when the backward leaves the block, and the grouped
        # function's backward -- all layers at once -- only runs when the LAST conditional batch norm has delivered its gradient: the layers keep their own launches
        return
    ys, yidx = [], []
    for _, y in ok:
        for j, t in enumerate(ys):
            if t is y:
                yidx.append(j)
                break
        else:
            yidx.append(len(ys))
            ys.append(y)
    if any(t.shape != ys[0].shape for t in ys):
        return
    metas = [(m.gain._sg_rt, m.bias._sg_rt, m._ones2) for m, _ in ok]
    ws = []
    for m, _ in ok:
        ws += [m.gain.master_weight, m.bias.master_weight]
    outs = CbnAffineGroupFn.apply(slot, metas, yidx, *ys, *ws)
    for (m, _), o in zip(ok, outs):
        slot.cbn_rows[id(m)] = o


_CBN_GROUP = [os.environ.get("SG_CBN_GROUP", "1") != "0"]      # all conditional batch norms' affine rows of a generator forward in one launch (CbnAffineGroupFn)


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


__all__ = ['CatConvDgradFn', 'CatConvFn', 'CbnAffineFn', 'CbnAffineGroupFn', 'cbn_prefetch', '_CBN_GROUP', 'ConvCfg', 'ConvDgradFn', 'ConvFn', 'ConvSkipFn', 'ConvTransposeFn', 'EmbeddingFn', 'GradLink', 'LinearDgradFn', 'LinearFn', 'SNEmbeddingFn', 'SliceUpFn', '_GRAD_LINK', '_QUAD', '_SKIP_FUSION', '_conv_dgrad', '_conv_fwd', '_conv_wgrad', '_quad_form']
