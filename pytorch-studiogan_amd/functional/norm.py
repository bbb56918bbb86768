"""(conditional) batch norm, synchronised across ranks (csrc/norm.hip, comm.hip, p2p.hip; reference src/utils/ops.py:14-28,227-228)."""
from ._base import *  # noqa: F401,F403  (shared helpers, switches, raw launch wrappers, torch / _lib / comm)

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
                partial = zeros_small(2 * Cc, torch.float64, dev)
                if fused is not None:
                    L.call("sg_bn_stats_from_tiles", fused[0].data_ptr(), fused[1], Cc, L.ptr(partial), L.stream())
                else:
                    L.call("sg_bn_partial_stats", L.dt(x), L.ptr(x), Cc, N * HW, Cc, L.ptr(partial), L.stream())
                count *= ws
                with _comm.exposed():
                    L.call("sg_bn_finalize_p2p", pp.handle, L.ptr(partial), count, Cc, cfg.eps, cfg.momentum, L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.stream())
            elif fused is not None:
                partial = zeros_small(2 * Cc, torch.float64, dev)
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
                partial = zeros_small(2 * Cc, torch.float64, dev)
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
        sums = zeros_small((N, Cc, 2), torch.float32, dev)
        bptr = (L.ptr(gain) + 4 * Cc) if cfg.packed else L.ptr(bias)       # packed cBN rows: [gain(C) | bias(C)], pitch gsn = 2 C
        L.call("sg_bn_bwd_reduce", L.dt(x), L.ptr(x), L.ptr(dy), N, HW, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(gain), bptr, gsn,
               1 if cfg.relu else 0, L.ptr(sums), L.stream())
        chan = torch.empty(2 * Cc, dtype=torch.float64, device=dev)
        dgain = zeros_like_small(gain) if (gain is not None and ctx.needs_input_grad[1]) else None
        dbias = zeros_like_small(bias) if (bias is not None and ctx.needs_input_grad[2]) else None
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
        sums = zeros_small((N, Cc, 2), torch.float32, dev)
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
        sums = zeros_small((N, Cc, 5), torch.float32, dev)
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
            dgain = zeros_like_small(gain)
            L.call("sg_bn_bwd2_dgain", L.ptr(chan_local), L.ptr(chan), ctx.count, L.ptr(invstd), Cc, use_batch, L.ptr(dgain), L.stream())
        return g_dy, g_x, dgain, None, None, None, None, None


__all__ = ['BNBwdFn', 'BNCfg', 'BNFn', '_allreduce_sum', '_world']
