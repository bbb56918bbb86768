"""GPU: every kernel family of libsgamd.so, called through the C ABI, against CPU torch (fp64/fp32) restatements of
the same op on the same seeded inputs. Tolerances: fp32 path 2e-4 of range (exact-fp32 MFMA; only summation order
differs), bf16 path 3e-2 (inputs are rounded to bf16 before BOTH computations, so what is measured is the kernel,
not the input quantisation; accumulation is fp32)."""
import pytest
import torch
import torch.nn.functional as TF

from util import check, tol_for

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(shape, generator=g) * scale).to(dtype)
    return x  # CPU tensor already rounded to the compute dtype


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("pf,qf", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("I,J,K,batch", [(256, 192, 128, 2), (100, 70, 52, 3), (96, 300, 64, 1), (16, 520, 40, 2)])
def test_gemm_forms(sg, dtype, pf, qf, I, J, K, batch):
    from studiogan_amd import functional as F, _lib as L
    P = rnd((batch, I, K) if pf == 0 else (batch, K, I), dtype, 1)
    Q = rnd((batch, J, K) if qf == 0 else (batch, K, J), dtype, 2)
    bias = rnd((I,), torch.float32, 3)
    Pm = P.double() if pf == 0 else P.double().transpose(1, 2)
    Qm = Q.double() if qf == 0 else Q.double().transpose(1, 2)
    ref = torch.einsum("bik,bjk->bji", Pm, Qm) * 0.5 + bias.double()
    d = dev()
    out = torch.empty((batch, J, I), dtype=torch.float32, device=d)
    Pd, Qd, bd = P.to(d), Q.to(d), bias.to(d)
    for no_tr in ([0, 1] if dtype == torch.bfloat16 and (pf or qf) else [0]):
        out.zero_()
        F.gemm_raw(L.dt(dtype), Pd, pf, P.shape[2], Qd, qf, Q.shape[2], out, I, I, J, K, batch=batch, p_bs=P.shape[1] * P.shape[2],
                   q_bs=Q.shape[1] * Q.shape[2], out_bs=J * I, bias=bd, alpha=0.5, epi_flags=L.EPI_OUT_F32, no_tr=no_tr)
        torch.cuda.synchronize()
        check(f"gemm p{pf}q{qf} {I}x{J}x{K} b{batch} no_tr={no_tr}", out, ref, 2e-4 if dtype == torch.float32 else 2e-3)


def test_linear_group(sg):
    """sg_linear_group (csrc/linear_group.hip): several small fp32 linear layers of different widths on shared / separate inputs in one launch, against fp64
    (ragged row tiles, a batch that is not a multiple of the tile, a K that is not a multiple of the k-chunk, an item without a bias)"""
    import ctypes
    from studiogan_amd import _lib as L
    d = dev()
    B, K = 100, 148
    ys = [rnd((B, K), torch.float32, 10 + g).to(d) for g in range(2)]
    spec = [(3072, 0, True), (192, 0, True), (70, 1, False), (1, 1, True)]
    arr = (L.LinearItem * len(spec))()
    keep, refs, outs = [], [], []
    for i, (rows, g, has_bias) in enumerate(spec):
        w = rnd((rows, K), torch.float32, 20 + i).to(d)
        b = rnd((rows,), torch.float32, 30 + i).to(d) if has_bias else None
        ldo = rows + 8
        o = torch.full((B, ldo), 7.0, dtype=torch.float32, device=d)
        it = arr[i]
        it.w, it.y, it.bias, it.out, it.rows, it.K, it.ldy, it.ldo = w.data_ptr(), ys[g].data_ptr(), (b.data_ptr() if has_bias else None), o.data_ptr(), rows, K, K, ldo
        keep += [w, b]
        outs.append(o)
        refs.append(ys[g].double() @ w.double().t() + (b.double() if has_bias else 0.0))
    tab = torch.frombuffer(bytearray(arr), dtype=torch.uint8).to(d)
    L.call("sg_linear_group", tab.data_ptr(), arr, len(spec), B, L.stream())
    torch.cuda.synchronize()
    for i, (rows, g, has_bias) in enumerate(spec):
        check(f"linear_group item {i} rows {rows}", outs[i][:, :rows], refs[i], 2e-5)
        assert (outs[i][:, rows:] == 7.0).all()          # nothing written past an item's rows


def _conv_ref(x, w, stride, pad, relu_in=False, up=False, pool=False, bias=None, res=None):
    x = x.double()
    if relu_in:
        x = torch.relu(x)
    if up:
        x = TF.interpolate(x, scale_factor=2, mode="nearest")
    y = TF.conv2d(x, w.double(), None if bias is None else bias.double(), stride=stride, padding=pad)
    if pool:
        y = TF.avg_pool2d(y, 2)
    if res is not None:
        y = y + res.double()
    return y


CONV_CASES = [
    # N, Cin, Cout, H, W, R, S, stride, (ph, pw)
    (2, 32, 64, 8, 8, 3, 3, 1, (1, 1)),
    (3, 96, 96, 8, 8, 3, 3, 1, (1, 1)),
    (2, 48, 192, 4, 4, 1, 1, 1, (0, 0)),
    (2, 3, 96, 16, 16, 3, 3, 1, (1, 1)),
    (2, 96, 3, 16, 16, 3, 3, 1, (1, 1)),
    (1, 20, 12, 9, 7, 3, 3, 1, (1, 1)),
    (2, 16, 40, 9, 9, 1, 7, 1, (0, 3)),
    (2, 16, 24, 11, 11, 3, 3, 2, (0, 0)),
    (2, 8, 16, 8, 8, 4, 4, 2, (1, 1)),
    (1, 384, 128, 4, 4, 3, 3, 1, (1, 1)),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(sg, dtype, case):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, W, R, S, stride, (ph, pw) = case
    d = dev()
    x = rnd((N, Cin, H, W), dtype, 11)
    w = rnd((Cout, Cin, R, S), dtype, 12, 0.2)
    bias = rnd((Cout,), torch.float32, 13)
    tol = 2e-4 if dtype == torch.float32 else 4e-3
    # forward
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yref = TF.conv2d(xr, wr, bias.double(), stride=stride, padding=(ph, pw))
    w_fwd = w.permute(0, 2, 3, 1).contiguous().to(d)      # [Cout][R][S][Cin]
    xd = nhwc(x).to(d)
    y = F.conv2d_raw(xd, w_fwd.data_ptr(), Cin, Cout, R, S, stride, ph, pw, bias=bias.to(d))
    torch.cuda.synchronize()
    check(f"conv fwd {case}", nchw(y.float().cpu()), yref, tol)
    # backward references
    gy = rnd(tuple(yref.shape), dtype, 14)
    yref.backward(gy.double())
    gyd = nhwc(gy).to(d)
    Ho, Wo = yref.shape[2], yref.shape[3]
    if stride == 1:
        w_dg = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(d)  # [Cin][R'][S'][Cout]
        dx = F.conv2d_raw(gyd, w_dg.data_ptr(), Cout, Cin, R, S, 1, R - 1 - ph, S - 1 - pw)
        torch.cuda.synchronize()
        check(f"conv dgrad {case}", nchw(dx.float().cpu()), xr.grad, tol)
    else:
        # strided convolution: data gradient = transposed gather over the UNflipped [Cin][R][S][Cout] image
        w_dg = w.permute(1, 2, 3, 0).contiguous().to(d)
        dx = F.conv2d_raw(gyd, w_dg.data_ptr(), Cout, Cin, R, S, stride, ph, pw, L.PIX_TRANSPOSED, transposed_out_hw=(H, W))
        torch.cuda.synchronize()
        check(f"conv strided dgrad {case}", nchw(dx.float().cpu()), xr.grad, tol)
    for no_tr in ([0, 1] if dtype == torch.bfloat16 else [0]):
        dw = torch.zeros((Cout, R, S, Cin), dtype=torch.float32, device=d)
        F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, S, Ho, Wo, stride, ph, pw, no_tr=no_tr)
        torch.cuda.synchronize()
        check(f"conv wgrad {case} no_tr={no_tr}", dw.cpu().permute(0, 3, 1, 2), wr.grad, tol)


def f32_split_case(d, case, sync=lambda: torch.cuda.synchronize()):
    """fp32 forward convolution in the two arithmetic modes of the generic engine against F.conv2d in fp64: exact (fp32 MFMA) and "bf16x3" (operands split into
    two bf16 terms in registers, three bf16 MFMAs per k-tile, csrc/gemm_core.h SPLIT). Returns (error exact, error bf16x3), both relative to the largest output."""
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, W, R, S, stride, (ph, pw) = case
    x = rnd((N, Cin, H, W), torch.float32, 21)
    w = rnd((Cout, Cin, R, S), torch.float32, 22, 0.2)
    bias = rnd((Cout,), torch.float32, 23)
    yref = TF.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=(ph, pw))
    w_fwd = w.permute(0, 2, 3, 1).contiguous().to(d)
    xd = nhwc(x).to(d)
    errs = []
    for mode in ("exact", "bf16x3"):
        with F.f32_mode(mode):
            assert L.lib().sg_get_f32_mode() == F.f32_mode.MODES[mode]
            y = F.conv2d_raw(xd, w_fwd.data_ptr(), Cin, Cout, R, S, stride, ph, pw, 0, L.EPI_RELU, bias=bias.to(d))
        sync()
        assert L.lib().sg_get_f32_mode() == 0
        errs.append(float((nchw(y.float().cpu()).double() - torch.relu(yref)).abs().max() / yref.abs().max()))
    return errs


def f32_split_wgrad_case(d, case, sync=lambda: torch.cuda.synchronize()):
    """fp32 weight gradient (reduction over pixels: [k][row] LDS images, csrc/gemm_core.h frag_mc_f32_split) in the two modes against autograd in fp64"""
    from studiogan_amd import functional as F
    N, Cin, Cout, H, W, R, S, stride, (ph, pw) = case
    x = rnd((N, Cin, H, W), torch.float32, 31)
    w = rnd((Cout, Cin, R, S), torch.float32, 32, 0.2).double().requires_grad_(True)
    y = TF.conv2d(x.double(), w, None, stride=stride, padding=(ph, pw))
    gy = rnd(tuple(y.shape), torch.float32, 33)
    y.backward(gy.double())
    xd, gyd = nhwc(x).to(d), nhwc(gy).to(d)
    errs = []
    for mode in ("exact", "bf16x3"):
        dw = torch.zeros((Cout, R, S, Cin), dtype=torch.float32, device=d)
        with F.f32_mode(mode):
            F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, S, y.shape[2], y.shape[3], stride, ph, pw)
        sync()
        errs.append(float((dw.cpu().permute(0, 3, 1, 2).double() - w.grad).abs().max() / w.grad.abs().max()))
    return errs


# Inception-like shapes on the all-vector path: 1x1 / 3x3 / 1x7 / 7x1 / 5x5, stride 2, couts that pad the 128 / 96 / 32-wide tiles, K from 64 to 3456
F32_SPLIT_CASES = [(2, 64, 96, 17, 17, 3, 3, 1, (1, 1)), (2, 192, 32, 9, 9, 1, 1, 1, (0, 0)), (1, 128, 160, 17, 17, 1, 7, 1, (0, 3)), (1, 160, 192, 17, 17, 7, 1, 1, (3, 0)),
                   (2, 48, 64, 13, 13, 5, 5, 1, (2, 2)), (2, 288, 384, 17, 17, 3, 3, 2, (0, 0)), (1, 384, 384, 8, 8, 3, 3, 1, (1, 1))]


@pytest.mark.parametrize("case", F32_SPLIT_CASES)
def test_conv_fwd_f32_bf16x3_split(sg, case):
    """bound: 2e-5 of the largest output for the split mode (measured <= 6e-6), 2e-6 for the exact one -- and the split must not be the exact kernel in disguise"""
    e_exact, e_split = f32_split_case(dev(), case)
    print(f"{case}: exact {e_exact:.2e}  bf16x3 {e_split:.2e}")
    assert e_exact <= 2e-6 and e_split <= 2e-5, (e_exact, e_split)
    assert e_split > e_exact
    w_exact, w_split = f32_split_wgrad_case(dev(), case)
    print(f"{case}: weight gradient exact {w_exact:.2e}  bf16x3 {w_split:.2e}")
    assert w_exact <= 4e-6 and w_split <= 2e-5 and w_split > w_exact, (w_exact, w_split)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(2, 16, 24, 4, 4, 4, 2, 1), (3, 64, 32, 8, 8, 4, 2, 1), (2, 8, 8, 5, 6, 3, 1, 1), (1, 12, 20, 3, 3, 5, 3, 2)])
def test_conv_transpose(sg, dtype, case):
    """nn.ConvTranspose2d forward / data gradient / weight gradient (reference utils/ops.py:176-184; DCGAN generator
    models/deep_conv.py:21) against F.conv_transpose2d in fp64."""
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, W, R, stride, pad = case
    d = dev()
    x = rnd((N, Cin, H, W), dtype, 21)
    w = rnd((Cin, Cout, R, R), dtype, 22, 0.2)              # torch's ConvTranspose2d weight layout
    bias = rnd((Cout,), torch.float32, 23)
    tol = 2e-4 if dtype == torch.float32 else 4e-3
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yref = TF.conv_transpose2d(xr, wr, bias.double(), stride=stride, padding=pad)
    Ho, Wo = yref.shape[2], yref.shape[3]
    w_fwd = w.permute(1, 2, 3, 0).contiguous().to(d)        # [Cout][R][S][Cin], unflipped
    xd = nhwc(x).to(d)
    y = F.conv2d_raw(xd, w_fwd.data_ptr(), Cin, Cout, R, R, stride, pad, pad, L.PIX_TRANSPOSED, bias=bias.to(d), transposed_out_hw=(Ho, Wo))
    torch.cuda.synchronize()
    check(f"deconv fwd {case}", nchw(y.float().cpu()), yref, tol)
    gy = rnd(tuple(yref.shape), dtype, 24)
    yref.backward(gy.double())
    gyd = nhwc(gy).to(d)
    w_dg = w.permute(0, 2, 3, 1).contiguous().to(d)         # [Cin][R][S][Cout]
    dx = F.conv2d_raw(gyd, w_dg.data_ptr(), Cout, Cin, R, R, stride, pad, pad)
    torch.cuda.synchronize()
    check(f"deconv dgrad {case}", nchw(dx.float().cpu()), xr.grad, tol)
    dw = torch.zeros((Cin, R, R, Cout), dtype=torch.float32, device=d)   # roles of x and dy exchanged
    F.conv2d_wgrad_raw(gyd, xd, dw.data_ptr(), Cout, Cin, R, R, H, W, stride, pad, pad)
    torch.cuda.synchronize()
    check(f"deconv wgrad {case}", dw.cpu().permute(0, 3, 1, 2), wr.grad, tol)


@pytest.mark.parametrize("dtype", DT)
def test_conv_fused_flags(sg, dtype):
    """ReLU-on-load, nearest x2 upsample-on-load, fused 2x2 average pooling, residual add, ReLU mask, and the matching
    backward views (pooled-gradient broadcast, pooling-sum of the upsample) -- the GenBlock / DiscBlock fusions."""
    from studiogan_amd import functional as F, _lib as L
    d = dev()
    N, Cin, Cout, H, W = 2, 32, 96, 8, 8
    tol = 2e-4 if dtype == torch.float32 else 4e-3
    x = rnd((N, Cin, H, W), dtype, 21)
    w = rnd((Cout, Cin, 3, 3), dtype, 22, 0.2)
    w_fwd = w.permute(0, 2, 3, 1).contiguous().to(d)
    w_dg = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(d)
    xd = nhwc(x).to(d)
    for relu_in, up, pool in [(True, False, True), (False, True, False), (True, True, True), (True, False, False)]:
        Ho = H * (2 if up else 1)
        Hy = Ho // 2 if pool else Ho
        res = rnd((N, Cout, Hy, Hy), dtype, 23)
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        yref = _conv_ref(xr, wr, 1, 1, relu_in, up, pool, None, res)
        pf = (L.PIX_RELU if relu_in else 0) | (L.PIX_UPSAMPLE if up else 0)
        ef = L.EPI_POOL if pool else 0
        y = F.conv2d_raw(xd, w_fwd.data_ptr(), Cin, Cout, 3, 3, 1, 1, 1, pf, ef, res=nhwc(res).to(d), alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        tag = f"relu={relu_in} up={up} pool={pool}"
        check("conv fused fwd " + tag, nchw(y.float().cpu()), yref, tol)
        gy = rnd(tuple(yref.shape), dtype, 24)
        yref.backward(gy.double())
        gyd = nhwc(gy).to(d)
        dx = F.conv2d_raw(gyd, w_dg.data_ptr(), Cout, Cin, 3, 3, 1, 1, 1, L.PIX_UPSAMPLE if pool else 0, L.EPI_POOL if up else 0,
                          mask=xd if relu_in else None, alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        check("conv fused dgrad " + tag, nchw(dx.float().cpu()), xr.grad, tol)
        dw = torch.zeros((Cout, 3, 3, Cin), dtype=torch.float32, device=d)
        F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, 3, 3, Ho, Ho, 1, 1, 1, pf, L.PIX_UPSAMPLE if pool else 0, alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        check("conv fused wgrad " + tag, dw.cpu().permute(0, 3, 1, 2), wr.grad, tol)


@pytest.mark.parametrize("dtype", DT)
def test_conv_large_splitk_and_tiles(sg, dtype):
    """A BigGAN-sized layer slice (many k-splits, 96-wide tile config, XCD remap with a non-multiple-of-8 tile count)."""
    from studiogan_amd import functional as F
    d = dev()
    N, Cin, Cout, H = 4, 96, 192, 32
    tol = 2e-4 if dtype == torch.float32 else 5e-3
    x = rnd((N, Cin, H, H), dtype, 31)
    w = rnd((Cout, Cin, 3, 3), dtype, 32, 0.1)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yref = TF.conv2d(xr, wr, None, 1, 1)
    gy = rnd(tuple(yref.shape), dtype, 33)
    yref.backward(gy.float())
    xd, gyd = nhwc(x).to(d), nhwc(gy).to(d)
    y = F.conv2d_raw(xd, w.permute(0, 2, 3, 1).contiguous().to(d).data_ptr(), Cin, Cout, 3, 3, 1, 1, 1)
    dw = torch.zeros((Cout, 3, 3, Cin), dtype=torch.float32, device=d)
    F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, 3, 3, H, H, 1, 1, 1)
    torch.cuda.synchronize()
    check("conv fwd large", nchw(y.float().cpu()), yref, tol)
    check("conv wgrad large", dw.cpu().permute(0, 3, 1, 2), wr.grad, tol)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mode", ["cbn_relu", "affine", "plain_eval"])
def test_batchnorm(sg, dtype, mode):
    from studiogan_amd import functional as F
    d = dev()
    N, Cc, H = 4, 24, 6
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    x = rnd((N, Cc, H, H), dtype, 41) * 2 + 0.5
    x = x.to(dtype)
    rm, rv = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    xr = x.double().requires_grad_(True)
    if mode == "cbn_relu":
        gain = (1 + 0.3 * torch.randn(N, Cc)).requires_grad_(True)
        bias = (0.3 * torch.randn(N, Cc)).requires_grad_(True)
        rm2, rv2 = rm.clone().double(), rv.clone().double()
        bn = TF.batch_norm(xr, rm2, rv2, None, None, True, 0.1, 1e-4)
        yref = torch.relu(bn * gain.double().view(N, Cc, 1, 1) + bias.double().view(N, Cc, 1, 1))
        cfg = F.BNCfg(True, True, 1e-4, 0.1, True)
    elif mode == "affine":
        gain = (1 + 0.3 * torch.randn(Cc)).requires_grad_(True)
        bias = (0.3 * torch.randn(Cc)).requires_grad_(True)
        rm2, rv2 = rm.clone().double(), rv.clone().double()
        yref = TF.batch_norm(xr, rm2, rv2, gain.double(), bias.double(), True, 0.1, 1e-4)
        cfg = F.BNCfg(True, True, 1e-4, 0.1, False)
    else:
        gain = bias = None
        rm2, rv2 = rm.clone().double(), rv.clone().double()
        yref = TF.batch_norm(xr, rm2, rv2, None, None, False, 0.1, 1e-4)
        cfg = F.BNCfg(False, False, 1e-4, 0.1, False)
    gy = rnd(tuple(yref.shape), dtype, 42)
    yref.backward(gy.double())
    xd = nhwc(x).to(d).requires_grad_(True)
    gd = gain.detach().float().to(d).requires_grad_(True) if gain is not None else None
    bd = bias.detach().float().to(d).requires_grad_(True) if bias is not None else None
    rmd, rvd = rm.clone().to(d), rv.clone().to(d)
    y = F.BNFn.apply(xd, gd, bd, rmd, rvd, cfg)
    y.backward(nhwc(gy).to(d))
    torch.cuda.synchronize()
    check(f"bn {mode} fwd", nchw(y.detach().float().cpu()), yref, tol)
    check(f"bn {mode} dx", nchw(xd.grad.float().cpu()), xr.grad, tol)
    if gain is not None:
        check(f"bn {mode} dgain", gd.grad.cpu(), gain.grad, tol)
        check(f"bn {mode} dbias", bd.grad.cpu(), bias.grad, tol)
    if mode != "plain_eval":
        check(f"bn {mode} running_mean", rmd.cpu(), rm2, tol)
        check(f"bn {mode} running_var", rvd.cpu(), rv2, tol)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(4, 24, 6), (3, 64, 16), (2, 40, 5)])      # scalar / streaming / vector kernels of sg_bn_bwd_apply_res
def test_batchnorm_backward_adds_the_linked_skip_gradient(sg, dtype, shape):
    """functional.GradLink on the BN side (generator blocks): the gradient the block input receives through the skip path is added INSIDE
    sg_bn_bwd_apply_res; dx with a link == dx without it + the handed-over tensor, on every kernel variant."""
    from studiogan_amd import functional as F
    d = dev()
    N, Cc, H = shape
    x = nhwc((rnd((N, Cc, H, H), dtype, 61) * 2 + 0.5).to(dtype)).to(d)
    gy = nhwc(rnd((N, Cc, H, H), dtype, 62)).to(d)
    r = nhwc(rnd((N, Cc, H, H), dtype, 63)).to(d).contiguous()
    gain = (1 + 0.3 * torch.randn(N, Cc)).to(d)
    bias = (0.3 * torch.randn(N, Cc)).to(d)
    cfg = F.BNCfg(True, False, 1e-4, 0.1, True)
    grads = []
    for use_link in (False, True):
        xd = x.clone().requires_grad_(True)
        link = F.GradLink()
        y = F.BNFn.apply(xd, gain, bias, None, None, cfg, link)
        if use_link:
            link.dx = r
        y.backward(gy)
        assert link.dx is None, "the backward must consume the handed-over gradient"
        grads.append(xd.grad.float().cpu())
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    check(f"bn backward + linked residual {dtype} {shape}", grads[1], grads[0].double() + r.float().cpu().double(), tol)


@pytest.mark.parametrize("dtype", DT)
def test_softmax_pool_misc(sg, dtype):
    from studiogan_amd import functional as F, _lib as L
    d = dev()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    # softmax fwd / bwd
    S = torch.randn(37, 200) * 3
    P = torch.softmax(S.double(), -1)
    Sd = S.to(d)
    Pd = torch.empty((37, 200), dtype=dtype, device=d)
    L.call("sg_softmax_rows", L.dt(dtype), Sd.data_ptr(), Pd.data_ptr(), 37, 200, L.stream())
    check("softmax fwd", Pd.float().cpu(), P, tol)
    dP = torch.randn(37, 200)
    Pq = Pd.float().cpu().double()
    dS_ref = Pq * (dP.double() - (dP.double() * Pq).sum(-1, keepdim=True))
    dSd = torch.empty((37, 200), dtype=dtype, device=d)
    L.call("sg_softmax_rows_bwd", L.dt(dtype), Pd.data_ptr(), dP.to(d).data_ptr(), dSd.data_ptr(), 37, 200, L.stream())
    check("softmax bwd", dSd.float().cpu(), dS_ref, tol)
    # avg / max pooling through autograd
    x = rnd((2, 12, 8, 8), dtype, 51)
    xr = x.double().requires_grad_(True)
    xd = nhwc(x).to(d).requires_grad_(True)
    y = F.AvgPool2Fn.apply(xd)
    yr = TF.avg_pool2d(xr, 2)
    gy = rnd(tuple(yr.shape), dtype, 52)
    y.backward(nhwc(gy).to(d)); yr.backward(gy.double())
    check("avgpool fwd", nchw(y.detach().float().cpu()), yr, tol)
    check("avgpool bwd", nchw(xd.grad.float().cpu()), xr.grad, tol)
    # add_relu
    a = rnd((2, 8, 4, 4), dtype, 53)
    ar, xr2 = a.double().requires_grad_(True), x[:, :8, :4, :4].double().requires_grad_(True)
    ad, xd2 = a.to(d).requires_grad_(True), x[:, :8, :4, :4].contiguous().to(d).requires_grad_(True)
    o = F.AddReluFn.apply(ad, xd2)
    orf = ar + torch.relu(xr2)
    g = rnd(tuple(orf.shape), dtype, 54)
    o.backward(g.to(d)); orf.backward(g.double())
    check("add_relu fwd", o.detach().float().cpu(), orf, tol)
    check("add_relu dx", xd2.grad.float().cpu(), xr2.grad, tol)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 4, 16), (3, 32, 32, 24, 24, 96), (2, 32, 32, 16, 12, 48), (2, 64, 32, 32, 32, 40), (1, 128, 128, 32, 32, 128)])
def test_attention_core(sg, dtype, shape, monkeypatch):
    """AttnCoreFn = maxpool + QK^T + softmax + PV and its backward vs the reference formulation (utils/ops.py:83-100).
    The 32x32 / 64x32 shapes take the fused score kernels of csrc/attn.hip in bf16 (G: 24 -> 96 channels, D: 12(16) -> 48); the 128 x 128 one is
    BigGAN-deep-256's discriminator attention (16384 queries x 4096 keys, 32 -> 128 channels; reference src/models/big_resnet_deep_legacy.py:80-95):
    keys and values streamed through LDS, forward and both backward kernels -- no score matrix in HBM."""
    from studiogan_amd import functional as F, _lib as L
    d = dev()
    B, H, W, Dp, Dv, Cg = shape
    if H * W > 8192:
        if dtype == torch.float32:
            pytest.skip("the 16384-query shape is the bf16 streaming path's case")
        assert L.lib().sg_attn_fwd_flash_ok(B, H * W, H * W // 4, Dp, Cg) == 1 and L.lib().sg_attn_bwd_fused_ok(B, H * W, H * W // 4, Dp, Cg) == 1
        calls = []
        orig = L.call
        monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
    tol = 5e-4 if dtype == torch.float32 else 3e-2
    th = rnd((B, Dp, H, W), dtype, 61, 0.5); th[:, Dv:] = 0
    ph = rnd((B, Dp, H, W), dtype, 62, 0.5); ph[:, Dv:] = 0
    g = rnd((B, Cg, H, W), dtype, 63)
    thr, phr, gr = [t.double().requires_grad_(True) for t in (th, ph, g)]
    theta = thr.view(B, Dp, H * W)
    phi = TF.max_pool2d(phr, 2, 2).view(B, Dp, H * W // 4)
    attn = torch.softmax(torch.bmm(theta.permute(0, 2, 1), phi), -1)
    gg = TF.max_pool2d(gr, 2, 2).view(B, Cg, H * W // 4)
    oref = torch.bmm(gg, attn.permute(0, 2, 1)).view(B, Cg, H, W)
    go = rnd(tuple(oref.shape), dtype, 64)
    oref.backward(go.double())
    thd, phd, gd = [nhwc(t).to(d).requires_grad_(True) for t in (th, ph, g)]
    o = F.AttnCoreFn.apply(thd, phd, gd)
    o.backward(nhwc(go).to(d))
    torch.cuda.synchronize()
    check("attn core fwd", nchw(o.detach().float().cpu()), oref, tol)
    check("attn dtheta", nchw(thd.grad.float().cpu())[:, :Dv], thr.grad[:, :Dv], tol)
    check("attn dphi", nchw(phd.grad.float().cpu())[:, :Dv], phr.grad[:, :Dv], tol)
    check("attn dg", nchw(gd.grad.float().cpu()), gr.grad, tol)
    if H * W > 8192:
        assert "sg_attn_fwd_fused" in calls and "sg_attn_bwd_fused" in calls and "sg_gemm" not in calls and "sg_softmax_rows" not in calls, calls


@pytest.mark.parametrize("shape", [(2, 32, 16, 48, 0.5), (1, 32, 32, 96, 3.0), (1, 64, 32, 40, 6.0)])
def test_attention_flash_single_pass_deferred_rescale(sg, shape):
    """k_attn_fwd_flash (round 5: ONE pass over the keys, the running maximum raised -- and l / O' rescaled -- only when a block exceeds it by more than e^8):
    score ranges from +-8 to +-2400 with a few much larger keys late in the sequence, so that the rescale fires in the middle of the pass; output against the
    fp64 softmax of the reference formulation (utils/ops.py:83-100)."""
    from studiogan_amd import functional as F
    B, H, Dp, Cg, scale = shape
    g0 = torch.Generator().manual_seed(7)
    th = (scale * torch.randn(B, H, H, Dp, generator=g0)).to(torch.bfloat16)
    ph = (scale * torch.randn(B, H, H, Dp, generator=g0)).to(torch.bfloat16)
    g = torch.randn(B, H, H, Cg, generator=g0).to(torch.bfloat16)
    ph[:, H - 2, H - 3] *= 4.0
    d = dev()
    with torch.no_grad():
        o = F.AttnCoreFn.apply(th.to(d), ph.to(d), g.to(d))
    torch.cuda.synchronize()
    t = th.double().reshape(B, H * H, Dp)
    p = TF.max_pool2d(ph.double().permute(0, 3, 1, 2), 2, 2).reshape(B, Dp, -1)
    gg = TF.max_pool2d(g.double().permute(0, 3, 1, 2), 2, 2).reshape(B, Cg, -1)
    ref = torch.bmm(torch.softmax(torch.bmm(t, p), -1), gg.permute(0, 2, 1)).reshape(B, H, H, Cg)
    assert bool(torch.isfinite(o).all())
    check(f"single-pass flash attention {shape}", o.float().cpu(), ref, 1e-2)


def test_head_losses_embedding(sg):
    from studiogan_amd import functional as F, _lib as L
    d = dev()
    # losses
    r, f = torch.randn(19), torch.randn(19)
    for kind, name in [(0, "hinge"), (1, "wasserstein"), (2, "vanilla")]:
        rr, fr = r.double().requires_grad_(True), f.double().requires_grad_(True)
        from oracle import restate as O
        lref = O.d_loss(name, rr, fr); lref.backward()
        rd, fd = r.to(d).requires_grad_(True), f.to(d).requires_grad_(True)
        l = F.DLossFn.apply(rd, fd, kind); (l * 0.5).backward()
        check(f"d_loss {name}", l.detach().cpu(), lref, 1e-5)
        check(f"d_loss {name} d_real", rd.grad.cpu() * 2, rr.grad, 1e-5)
        check(f"d_loss {name} d_fake", fd.grad.cpu() * 2, fr.grad, 1e-5)
        fr2 = f.double().requires_grad_(True)
        gref = O.g_loss(name, fr2); gref.backward()
        fd2 = f.to(d).requires_grad_(True)
        gl = F.GLossFn.apply(fd2, kind); gl.backward()
        check(f"g_loss {name}", gl.detach().cpu(), gref, 1e-5)
        check(f"g_loss {name} d_fake", fd2.grad.cpu(), fr2.grad, 1e-5)
    # relu-sum over HW
    x = torch.randn(3, 4, 4, 20)
    xr = x.double().requires_grad_(True)
    href = torch.relu(xr).sum(dim=[1, 2])
    xd = x.to(d).requires_grad_(True)
    h = F.ReluSumFn.apply(xd)
    gh = torch.randn(3, 20)
    h.backward(gh.to(d)); href.backward(gh.double())
    check("relu_sum fwd", h.detach().cpu(), href, 1e-5)
    check("relu_sum bwd", xd.grad.cpu(), xr.grad, 1e-5)
    # plain embedding
    tab = torch.randn(10, 16)
    idx = torch.tensor([1, 3, 3, 9, 0])
    tr = tab.double().requires_grad_(True)
    eref = TF.embedding(idx, tr)
    td = torch.nn.Parameter(tab.to(d))
    e = F.EmbeddingFn.apply(td, idx.to(d))
    ge = torch.randn(5, 16)
    e.backward(ge.to(d)); eref.backward(ge.double())
    check("embedding fwd", e.detach().cpu(), eref, 1e-6)
    check("embedding bwd", td.grad.cpu(), tr.grad, 1e-6)


def test_adam_ema_and_lerp(sg):
    from studiogan_amd import _lib as L
    d = dev()
    n = 10007
    p, g = torch.randn(n), torch.randn(n)
    ema0 = torch.randn(n)
    pr = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.5, 0.999), eps=1e-6)
    pd, md, vd, ed = p.to(d).contiguous(), torch.zeros(n, device=d), torch.zeros(n, device=d), ema0.to(d)
    er = ema0.clone()
    for t in range(1, 4):
        gt = g * t
        pr.grad = gt.clone()
        opt.step()
        er = pr.detach().lerp(er, 0.9)
        L.call("sg_adam_ema", pd.data_ptr(), gt.to(d).data_ptr(), md.data_ptr(), vd.data_ptr(), ed.data_ptr(), n, 2e-4, 0.5, 0.999, 1e-6, 0.0, t, 0.9, 1.0, L.stream())
    check("adam p", pd.cpu(), pr.detach(), 1e-6)
    check("adam ema", ed.cpu(), er, 1e-6)
    # beta1 = 0 (BigGAN) and decay 0 (before g_ema_start: hard copy)
    pr2 = torch.nn.Parameter(p.clone())
    opt2 = torch.optim.Adam([pr2], lr=5e-5, betas=(0.0, 0.999), eps=1e-6)
    pr2.grad = g.clone(); opt2.step()
    pd2, md2, vd2, ed2 = p.to(d).contiguous(), torch.zeros(n, device=d), torch.zeros(n, device=d), ema0.to(d)
    L.call("sg_adam_ema", pd2.data_ptr(), g.to(d).data_ptr(), md2.data_ptr(), vd2.data_ptr(), ed2.data_ptr(), n, 5e-5, 0.0, 0.999, 1e-6, 0.0, 1, 0.0, 1.0, L.stream())
    check("adam beta1=0", pd2.cpu(), pr2.detach(), 1e-6)
    assert torch.equal(ed2.cpu(), pd2.cpu()), "EMA with decay 0 must be a hard copy"


def test_fused_adam_resumes_from_torch_adam_checkpoint(sg):
    """A torch.optim.Adam checkpoint (what the reference's ckpt.py writes) loaded into FusedAdam: the NEXT step equals torch's next
    step -- moments and bias-correction step count are restored, not restarted."""
    import copy
    from studiogan_amd.optim import FusedAdam
    d = dev()
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Linear(33, 65), torch.nn.Linear(65, 9))
    ropt = torch.optim.Adam(ref.parameters(), lr=2e-4, betas=(0.5, 0.999), eps=1e-6)
    for _ in range(4):
        ropt.zero_grad()
        ref(torch.randn(16, 33)).square().sum().backward()
        ropt.step()
    net = copy.deepcopy(ref).to(d)
    opt = FusedAdam(net.parameters(), lr=1.0, betas=(0.9, 0.9), eps=1e-6)
    opt.load_state_dict(copy.deepcopy(ropt.state_dict()))
    x = torch.randn(16, 33)
    ropt.zero_grad(); ref(x).square().sum().backward(); ropt.step()
    opt.zero_grad(); net(x.to(d)).square().sum().backward(); opt.step()
    torch.cuda.synchronize()
    for (k, a), b in zip(net.named_parameters(), ref.parameters()):
        check("resumed step " + k, a.detach().cpu(), b.detach(), 2e-6)
    sd = opt.state_dict()
    for i, q in enumerate(ref.parameters()):
        check(f"exp_avg[{i}]", sd["state"][i]["exp_avg"].cpu(), ropt.state[q]["exp_avg"], 1e-5)
        check(f"exp_avg_sq[{i}]", sd["state"][i]["exp_avg_sq"].cpu(), ropt.state[q]["exp_avg_sq"], 1e-5)
        assert float(sd["state"][i]["step"]) == 5.0


def test_quantize_bit_exact_and_resize(sg):
    """uint8 quantisation (utils/ops.py:251-255) must be bit-exact; the legacy bilinear resize (utils/resize.py:87-91)
    within fp32 tolerance."""
    from studiogan_amd import _lib as L
    import numpy as np
    d = dev()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(3, 3, 32, 32, generator=g) * 2.2 - 1.1
    # include exact grid points and half-way cases
    x[0, 0, 0, :] = torch.arange(32).float() / 127.5 - 1.0
    x[0, 0, 1, :] = (torch.arange(32).float() + 0.5) / 127.5 - 1.0
    xq = (x + 1) / 2
    xq = (255.0 * xq + 0.5).clamp(0.0, 255.0)
    q_ref = xq.numpy().astype(np.uint8)
    xd = x.to(d)
    out = torch.empty((3, 299, 299, 3), dtype=torch.float32, device=d)
    qd = torch.empty((3, 3, 32, 32), dtype=torch.uint8, device=d)
    L.call("sg_quantize_resize_normalize", 0, xd.data_ptr(), out.data_ptr(), qd.data_ptr(), 3, 3, 32, 32, 299, 299, 1, L.stream())
    assert np.array_equal(qd.cpu().numpy(), q_ref), "uint8 quantisation is not bit-exact"
    r = TF.interpolate(torch.from_numpy(q_ref).float(), size=(299, 299), mode="bilinear", align_corners=False).clamp(0, 255)
    r = (r / 255.0 - 0.5) / 0.5
    check("resize+normalize", out.cpu().permute(0, 3, 1, 2), r, 1e-5)


def test_no_cpu_fallback(sg):
    """The product path must fail loudly on CPU tensors (no silent eager fallback)."""
    from studiogan_amd import ops
    m = ops.snconv2d(4, 8, 3, 1, 1)
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 4, 8, 8))


@pytest.mark.parametrize("shape", [(3, 32, 32, 24, 96), (2, 32, 32, 16, 48), (2, 64, 32, 32, 40), (1, 64, 64, 16, 48)])
def test_attention_fused_forward_matches_unfused_chain(sg, shape):
    """sg_attn_fwd_fused (scores, softmax and P.V in one launch) against the chain it replaces (sg_attn_probs_fwd + batched GEMM) on the
    same inputs: identical probabilities / log-sum-exp, the output to fp32-summation-order differences of bf16 products; and without
    the probability store (no-grad forward)."""
    from studiogan_amd import functional as F, _lib as L
    d = dev()
    B, H, W, Dp, Cg = shape
    HW, HW4 = H * W, H * W // 4
    T = torch.bfloat16
    theta = rnd((B, HW, Dp), T, 71, 0.7).to(d)
    phi = rnd((B, HW4, Dp), T, 72, 0.7).to(d)
    g = rnd((B, HW4, Cg), T, 73).to(d)
    assert L.lib().sg_attn_fwd_fused_ok(B, HW, HW4, Dp, Cg) == 1
    P0 = torch.empty((B, HW, HW4), dtype=T, device=d)
    lse0 = torch.empty((B, HW), dtype=torch.float32, device=d)
    L.call("sg_attn_probs_fwd", L.ptr(theta), L.ptr(phi), L.ptr(P0), L.ptr(lse0), B, HW, HW4, Dp, L.stream())
    o0 = torch.empty((B, HW, Cg), dtype=T, device=d)
    F.gemm_raw(L.BF16, g, 1, Cg, P0, 0, HW4, o0, Cg, Cg, HW, HW4, batch=B, p_bs=HW4 * Cg, q_bs=HW * HW4, out_bs=HW * Cg)
    for store in (True, False):
        P1 = torch.zeros((B, HW, HW4), dtype=T, device=d) if store else None
        lse1 = torch.empty((B, HW), dtype=torch.float32, device=d)
        o1 = torch.empty((B, HW, Cg), dtype=T, device=d)
        o32 = None if store else torch.empty((B, HW, Cg), dtype=torch.float32, device=d)
        L.call("sg_attn_fwd_fused", L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(P1), L.ptr(lse1), L.ptr(o1), L.ptr(o32), B, HW, HW4, Dp, Cg, L.stream())
        torch.cuda.synchronize()
        if o32 is not None:
            assert torch.equal(o32.to(T), o1), "the fp32 copy of O must round to the bf16 output"
        if store:
            assert torch.equal(P1, P0), "fused forward must store the same bf16 probabilities"
            assert torch.equal(lse1, lse0)
        else:   # max pass + unnormalised pass (k_attn_fwd_flash): same statistics up to the summation order / exp2 form
            assert (lse1 - lse0).abs().max().item() < 2e-5
        check(f"fused attention output (store_p={store})", o1.float().cpu(), o0.float().cpu(), 8e-3)
    # against fp64 softmax(theta phi^T) g
    ref = torch.softmax(theta.double().cpu() @ phi.double().cpu().transpose(1, 2), -1) @ g.double().cpu()
    check("fused attention vs fp64", o1.float().cpu(), ref, 3e-2)


@pytest.mark.parametrize("shape", [(3, 32, 32, 24, 96), (2, 32, 32, 16, 48), (2, 64, 32, 32, 40), (1, 64, 64, 16, 48)])
def test_attention_fused_backward_matches_unfused_chain(sg, shape):
    """sg_attn_bwd_fused (query side: delta, dS in registers, dtheta; key side: P and dS recomputed in the transposed orientation, dphi, dg)
    against the chain it replaces (stored P, sg_attn_ds_bwd + three batched GEMMs) and against fp64 autograd of softmax(theta phi^T) g."""
    from studiogan_amd import functional as F, _lib as L
    d = dev()
    B, H, W, Dp, Cg = shape
    HW, HW4 = H * W, H * W // 4
    T = torch.bfloat16
    theta = rnd((B, HW, Dp), T, 81, 0.7).to(d)
    phi = rnd((B, HW4, Dp), T, 82, 0.7).to(d)
    g = rnd((B, HW4, Cg), T, 83).to(d)
    do = rnd((B, HW, Cg), T, 84).to(d)
    assert L.lib().sg_attn_bwd_fused_ok(B, HW, HW4, Dp, Cg) == 1
    P = torch.empty((B, HW, HW4), dtype=T, device=d)
    lse = torch.empty((B, HW), dtype=torch.float32, device=d)
    L.call("sg_attn_probs_fwd", L.ptr(theta), L.ptr(phi), L.ptr(P), L.ptr(lse), B, HW, HW4, Dp, L.stream())
    sd = L.BF16
    dg0 = torch.empty((B, HW4, Cg), dtype=T, device=d)
    F.gemm_raw(sd, do, 1, Cg, P, 1, HW4, dg0, Cg, Cg, HW4, HW, batch=B, p_bs=HW * Cg, q_bs=HW * HW4, out_bs=HW4 * Cg)
    dS = torch.empty((B, HW, HW4), dtype=T, device=d)
    L.call("sg_attn_ds_bwd", L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(do), L.ptr(lse), L.ptr(dS), B, HW, HW4, Dp, Cg, L.stream())
    dth0 = torch.empty((B, HW, Dp), dtype=T, device=d)
    F.gemm_raw(sd, phi, 1, Dp, dS, 0, HW4, dth0, Dp, Dp, HW, HW4, batch=B, p_bs=HW4 * Dp, q_bs=HW * HW4, out_bs=HW * Dp)
    dph0 = torch.empty((B, HW4, Dp), dtype=T, device=d)
    F.gemm_raw(sd, theta, 1, Dp, dS, 1, HW4, dph0, Dp, Dp, HW4, HW, batch=B, p_bs=HW * Dp, q_bs=HW * HW4, out_bs=HW4 * Dp)
    tr, pr, gr = [t.double().cpu().requires_grad_(True) for t in (theta, phi, g)]
    o = torch.softmax(tr @ pr.transpose(1, 2), -1) @ gr
    o.backward(do.double().cpu())
    # the forward output as the product path keeps it for the backward (unrounded fp32 copy from the fused forward): delta_q = dO_q . O_q
    lse_f = torch.empty((B, HW), dtype=torch.float32, device=d)
    o_f = torch.empty((B, HW, Cg), dtype=T, device=d)
    o32 = torch.empty((B, HW, Cg), dtype=torch.float32, device=d)
    L.call("sg_attn_fwd_fused", L.ptr(theta), L.ptr(phi), L.ptr(g), None, L.ptr(lse_f), L.ptr(o_f), L.ptr(o32), B, HW, HW4, Dp, Cg, L.stream())
    for what, o_arg, lse_arg in (("delta from a key pass", None, lse), ("delta = dO . O", o32, lse_f)):
        delta = torch.empty((B, HW), dtype=torch.float32, device=d)
        dth1, dph1, dg1 = torch.empty_like(dth0), torch.empty_like(dph0), torch.empty_like(dg0)
        L.call("sg_attn_bwd_fused", L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(do), L.ptr(o_arg), L.ptr(lse_arg), L.ptr(delta), L.ptr(dth1), L.ptr(dph1),
               L.ptr(dg1), B, HW, HW4, Dp, Cg, L.stream())
        torch.cuda.synchronize()
        check(f"fused bwd dtheta vs chain ({what})", dth1.float().cpu(), dth0.float().cpu(), 1e-2)
        check(f"fused bwd dphi vs chain ({what})", dph1.float().cpu(), dph0.float().cpu(), 1e-2)
        check(f"fused bwd dg vs chain ({what})", dg1.float().cpu(), dg0.float().cpu(), 1e-2)
        check(f"fused bwd dtheta vs fp64 ({what})", dth1.float().cpu(), tr.grad, 3e-2)
        check(f"fused bwd dphi vs fp64 ({what})", dph1.float().cpu(), pr.grad, 3e-2)
        check(f"fused bwd dg vs fp64 ({what})", dg1.float().cpu(), gr.grad, 3e-2)
        dref = (do.double().cpu() * o).sum(-1)
        check(f"delta ({what})", delta.cpu().double(), dref, 5e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n", [1000, 65536, 8 * 123457, 3 * 1048576 + 8])      # scalar walk; 16-byte path with the 4-deep body / its tail
def test_dot_product(sg, dtype, n):
    """sg_dot (the attention gate's gradient <dy, conv1x1(o)> and the squared-norm statistics): out += scale * <x, y>, both code paths."""
    from studiogan_amd import _lib as L
    d = dev()
    x = torch.randn(n, generator=torch.Generator().manual_seed(5)).to(dtype)
    y = torch.randn(n, generator=torch.Generator().manual_seed(6)).to(dtype)
    ref = 0.5 + 0.25 * float((x.double() * y.double()).sum())
    xd, yd = x.to(d), y.to(d)
    out = torch.full((1,), 0.5, dtype=torch.float32, device=d)
    L.call("sg_dot", L.dt(dtype), xd.data_ptr(), yd.data_ptr(), n, out.data_ptr(), 0.25, None, L.stream())
    torch.cuda.synchronize()
    assert abs(float(out) - ref) <= 2e-4 * (abs(ref) + n ** 0.5), (float(out), ref)


@pytest.mark.parametrize("with_emb", [False, True])
@pytest.mark.parametrize("shape", [(5, 40), (256, 1536), (37, 96)])
def test_projection_head_backward(sg, shape, with_emb):
    """sg_pd_head_bwd: dh = dadv (w1 + emb), dw1 += sum_b dadv h, db1 += sum dadv, demb = dadv h (reference big_resnet.py:363,387 backward)."""
    from studiogan_amd import _lib as L
    d = dev()
    B, Cc = shape
    g = torch.Generator().manual_seed(11)
    h, w1, emb, dadv = torch.randn(B, Cc, generator=g), torch.randn(Cc, generator=g), torch.randn(B, Cc, generator=g), torch.randn(B, generator=g)
    dw0, db0 = torch.randn(Cc, generator=g), torch.randn(1, generator=g)
    hd, wd, ed, gd = h.to(d), w1.to(d), emb.to(d), dadv.to(d)
    dh = torch.empty(B, Cc, device=d)
    demb = torch.empty(B, Cc, device=d)
    dw1, db1 = dw0.to(d), db0.to(d)
    L.call("sg_pd_head_bwd", hd.data_ptr(), wd.data_ptr(), ed.data_ptr() if with_emb else None, gd.data_ptr(), dh.data_ptr(), dw1.data_ptr(),
           db1.data_ptr(), demb.data_ptr() if with_emb else None, B, Cc, L.stream())
    torch.cuda.synchronize()
    wt = w1.double()[None, :] + (emb.double() if with_emb else 0.0)
    check("pd head dh", dh.cpu(), dadv.double()[:, None] * wt, 1e-5)
    check("pd head dw1", dw1.cpu(), dw0.double() + (dadv.double()[:, None] * h.double()).sum(0), 1e-5)
    check("pd head db1", db1.cpu(), db0.double() + dadv.double().sum(), 1e-5)
    if with_emb:
        check("pd head demb", demb.cpu(), dadv.double()[:, None] * h.double(), 1e-5)
