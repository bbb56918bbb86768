"""GPU: the quad convolutions (csrc/conv_q.h, wgrad_q.h, conv_q.hip) -- avgpool2(conv3x3(x)) as a 4x4 / stride-2 convolution and
conv3x3(up2(x)) as four 2x2 phase convolutions -- against torch's own conv2d + avg_pool2d / interpolate on the CPU (fp32 accumulation of the
same bf16 inputs), forward, data gradient and weight gradient, with every fused flag, on shapes that select every chunk geometry of the
weight-gradient kernel. Replaces src/models/big_resnet.py:28-42 (F.interpolate + conv2d1) and :177-192,221-242 (conv2d2 + average_pooling)."""
import os

import pytest
import torch

import quad_ref as Q
from util import check
from test_kernels_gpu import rnd

pytestmark = pytest.mark.gpu


def _dev(t):
    return t.to(torch.device("cuda:0"))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_quad_pack_matches_reference(sg, mode):
    from studiogan_amd import functional as F
    M, Cs = 96, 64
    for dt in (torch.bfloat16, torch.float32):
        w = rnd((M, 3, 3, Cs), dt, 300 + mode, 0.1)
        dst = torch.empty(M, 16, Cs, dtype=dt, device="cuda:0")
        wd = _dev(w)
        F.quad_pack_raw(wd.data_ptr(), dst, mode, M, Cs)
        torch.cuda.synchronize()
        ref = Q.quad_pack_ref(w.double(), mode).reshape(M, 16, Cs)
        check(f"quad pack mode {mode} {dt}", dst.float().cpu(), ref, 4e-3 if dt == torch.bfloat16 else 1e-6)


FWD_CASES = [
    # form, N, Hl, Wl, C, Cout, relu_in, bias, mask, res, relu_out
    (0, 2, 8, 8, 64, 96, False, True, False, False, False),
    (0, 2, 8, 8, 96, 96, True, True, False, False, False),       # D block tail: ReLU on load
    (0, 3, 4, 4, 32, 64, True, False, False, True, True),        # 4 x 4 low-resolution images, residual + ReLU on store, 64-cout tile
    (0, 1, 16, 64, 96, 192, True, True, False, False, False),    # wide rows, two cout tiles
    (0, 5, 2, 4, 32, 64, False, True, False, False, False),      # J = 40: partial tile, Hl = 2
    (0, 2, 32, 32, 64, 64, False, True, False, False, False),    # several pixel tiles
    (1, 2, 8, 8, 64, 96, False, True, False, False, False),
    (1, 2, 8, 8, 96, 96, False, True, True, False, False),       # data gradient of a D block tail: ReLU mask of the fine tensor
    (1, 3, 4, 4, 32, 64, False, False, True, True, False),       # mask AND residual (GradLink)
    (1, 1, 16, 64, 96, 192, False, True, False, False, False),
    (1, 5, 2, 4, 32, 64, True, True, False, False, True),
    (1, 2, 32, 32, 64, 64, False, False, False, True, False),
]


@pytest.mark.parametrize("bj", ["256", "128", "256db"])
@pytest.mark.parametrize("case", FWD_CASES)
def test_conv_q_matches_torch(sg, case, bj, monkeypatch):
    from studiogan_amd import functional as F, _lib as L
    monkeypatch.setenv("SG_CONV_Q_BJ", bj[:3])      # both pixel tiles (the launcher picks 128 when 256 leaves the chip under-filled)
    monkeypatch.setenv("SG_CONV_Q_DB", "1" if bj.endswith("db") else "0")      # and the double-buffered-patch variant of the 256 tile
    form, N, Hl, Wl, C, Cout, relu_in, with_bias, with_mask, with_res, relu_out = case
    dt = torch.bfloat16
    Hx, Wx = (2 * Hl, 2 * Wl) if form == 0 else (Hl, Wl)
    Hy, Wy = (Hl, Wl) if form == 0 else (2 * Hl, 2 * Wl)
    x = rnd((N, Hx, Wx, C), dt, 311)
    w9 = rnd((Cout, 3, 3, C), dt, 312, 0.1)
    bias = rnd((Cout,), torch.float32, 313) if with_bias else None
    mask = rnd((N, Hy, Wy, Cout), dt, 314) if with_mask else None
    res = rnd((N, Hy, Wy, Cout), dt, 315) if with_res else None
    xf, wf = x.float(), w9.float()
    ref = Q.pool_conv_torch(xf, wf, relu_in) if form == 0 else Q.up_conv_torch(xf, wf, relu_in)
    if bias is not None:
        ref = ref + bias
    if mask is not None:
        ref = ref * (mask.float() > 0)
    if res is not None:
        ref = ref + res.float()
    if relu_out:
        ref = torch.relu(ref)
    wq = torch.empty(Cout, 16, C, dtype=dt, device="cuda:0")
    w9d = _dev(w9)
    F.quad_pack_raw(w9d.data_ptr(), wq, form, Cout, C)
    y = F.conv2d_q_raw(_dev(x), wq.data_ptr(), form, C, Cout, L.PIX_RELU if relu_in else 0, L.EPI_RELU if relu_out else 0,
                       bias=None if bias is None else _dev(bias), res=None if res is None else _dev(res), mask=None if mask is None else _dev(mask))
    assert y is not None, "the quad kernel refused an eligible problem"
    torch.cuda.synchronize()
    check(f"conv_q {case}", y.float().cpu(), ref, 6e-3)
    # against the index-level restatement run on the kernel's OWN bf16 filter image: only accumulation order and the output rounding differ
    xx = torch.relu(xf) if relu_in else xf
    r2 = Q.convq_ref(xx.double(), wq.float().cpu().double().reshape(Cout, 4, 4, C), form)
    if bias is not None:
        r2 = r2 + bias.double()
    if mask is not None:
        r2 = r2 * (mask.double() > 0)
    if res is not None:
        r2 = r2 + res.double()
    if relu_out:
        r2 = torch.relu(r2)
    check(f"conv_q vs restatement {case}", y.float().cpu(), r2, 4e-3)


def test_conv_q_data_gradients_match_autograd(sg):
    """dgrad of POOL = UP form with the mode-2 image, dgrad of UP = POOL form with the mode-3 image (from the flipped transposed 3x3 image)"""
    from studiogan_amd import functional as F, _lib as L
    dt = torch.bfloat16
    N, Hl, Wl, C, Cout = 2, 8, 16, 64, 96
    w9 = rnd((Cout, 3, 3, C), dt, 321, 0.1)
    wft = Q.flipped_transposed(w9)                      # [C][3][3][Cout]
    wftd = _dev(wft)
    for form in (0, 1):
        Hx, Wx = (2 * Hl, 2 * Wl) if form == 0 else (Hl, Wl)
        x = rnd((N, Hx, Wx, C), dt, 322).float().requires_grad_(True)
        y = Q.pool_conv_torch(x, w9.float()) if form == 0 else Q.up_conv_torch(x, w9.float())
        dy = rnd(tuple(y.shape), dt, 323)
        (dx,) = torch.autograd.grad(y, x, dy.float())
        wq = torch.empty(C, 16, Cout, dtype=dt, device="cuda:0")
        F.quad_pack_raw(wftd.data_ptr(), wq, 2 + form, C, Cout)
        got = F.conv2d_q_raw(_dev(dy), wq.data_ptr(), 1 - form, Cout, C)
        assert got is not None
        torch.cuda.synchronize()
        check(f"conv_q dgrad of form {form}", got.float().cpu(), dx, 6e-3)


WG_CASES = [
    # form, N, Hl, Wl, C, Cout, relu, bias
    (0, 4, 4, 4, 64, 96, True, True),         # WC = 4: chunks of four 4 x 4 images, S = 2
    (1, 4, 4, 4, 32, 64, False, True),        # S = 1, NB = 2
    (0, 2, 8, 8, 96, 96, True, True),         # WC = 8, S = 1 (C % 64 != 0)
    (1, 2, 8, 8, 64, 192, False, True),       # two cout tiles
    (0, 1, 16, 16, 64, 64, False, False),     # WC = 16
    (1, 1, 16, 16, 128, 96, False, True),     # two input-channel groups
    (0, 1, 8, 32, 64, 96, True, True),        # WC = 32
    (1, 2, 4, 32, 32, 96, False, False),
    (0, 1, 4, 64, 64, 96, False, True),       # WC = 64
    (1, 1, 2, 128, 64, 64, False, True),      # two chunks per image row
]


@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad_q_matches_autograd(sg, case):
    from studiogan_amd import functional as F, _lib as L
    form, N, Hl, Wl, C, Cout, relu, with_bias = case
    dt = torch.bfloat16
    Hx, Wx = (2 * Hl, 2 * Wl) if form == 0 else (Hl, Wl)
    x = rnd((N, Hx, Wx, C), dt, 331)
    w9 = rnd((Cout, 3, 3, C), dt, 332, 0.1).float().requires_grad_(True)
    y = Q.pool_conv_torch(x.float(), w9, relu) if form == 0 else Q.up_conv_torch(x.float(), w9, relu)
    dy = rnd(tuple(y.shape), dt, 333)
    (dw,) = torch.autograd.grad(y, w9, dy.float())
    for splits in (0, 1, 3):
        dwd = torch.zeros(Cout, 9, C, dtype=torch.float32, device="cuda:0")
        dwd[0, 0, 0] = 1.0                              # the kernel ACCUMULATES
        db = torch.zeros(Cout, dtype=torch.float32, device="cuda:0") if with_bias else None
        ok = F.conv2d_q_wgrad_raw(_dev(x), _dev(dy), dwd.data_ptr(), form, C, Cout, L.PIX_RELU if relu else 0, dbias=db, splits=splits)
        assert ok, "the quad weight-gradient kernel refused an eligible problem"
        torch.cuda.synchronize()
        got = dwd.cpu().reshape(Cout, 3, 3, C)
        got[0, 0, 0, 0] -= 1.0
        check(f"wgrad_q {case} splits {splits}", got, dw, 3e-3)
        if with_bias:
            check(f"wgrad_q bias {case} splits {splits}", db.cpu(), dy.float().sum((0, 1, 2)), 2e-3)


@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad_q_lean_matches_round4_kernel(sg, case, monkeypatch):
    """csrc/wgrad_ql.h (the default since round 5) against the round-4 kernel it replaced (SG_WGRAD_Q_LEAN=0, csrc/wgrad_q.h): the same MFMAs in the same
    order -> the 3x3 gradient bit for bit, the bias gradient to fp32 rounding (summed through v_dot2 on other waves)."""
    from studiogan_amd import functional as F, _lib as L
    form, N, Hl, Wl, C, Cout, relu, with_bias = case
    dt = torch.bfloat16
    Hx, Wx = (2 * Hl, 2 * Wl) if form == 0 else (Hl, Wl)
    Hg, Wg = (Hl, Wl) if form == 0 else (2 * Hl, 2 * Wl)
    x, dy = _dev(rnd((N, Hx, Wx, C), dt, 341)), _dev(rnd((N, Hg, Wg, Cout), dt, 342))
    outs = {}
    for lean in ("0", "1"):
        monkeypatch.setenv("SG_WGRAD_Q_LEAN", lean)
        for splits in (0, 3):
            dwd = torch.zeros(Cout, 9, C, dtype=torch.float32, device="cuda:0")
            db = torch.zeros(Cout, dtype=torch.float32, device="cuda:0")
            assert F.conv2d_q_wgrad_raw(x, dy, dwd.data_ptr(), form, C, Cout, L.PIX_RELU if relu else 0, alpha=0.5, dbias=db, splits=splits)
            torch.cuda.synchronize()
            outs[(lean, splits)] = (dwd.cpu(), db.cpu())
    for splits in (0, 3):
        for lean in ("1",):
            assert torch.equal(outs[("0", splits)][0], outs[(lean, splits)][0]), (case, splits, lean)
            check(f"wgrad_q lean={lean} bias {case} splits {splits}", outs[(lean, splits)][1], outs[("0", splits)][1], 1e-5)


SKIP_CASES = [
    # N, Hl, Wl, C, Cout, C2, relu, bj
    (2, 8, 8, 64, 96, 32, True, "256"),
    (2, 8, 8, 96, 192, 96, True, "128"),
    (3, 4, 4, 64, 64, 64, False, "256"),
    (1, 16, 32, 192, 192, 96, True, "256"),
    (1, 16, 32, 192, 192, 96, True, "256db"),
    (2, 8, 64, 96, 96, 32, False, "256db"),
    (2, 16, 16, 96, 96, 8, True, "256"),          # the first discriminator block: the skip reads the 8-channel image (no ReLU on it): one slice = its four views
    (3, 4, 8, 64, 64, 8, True, "128"),
]


@pytest.mark.parametrize("case", SKIP_CASES)
def test_conv_q_fused_skip_matches_torch(sg, case, monkeypatch):
    """POOL form with the block's 1x1 skip in the same launch: avgpool2(conv3x3(relu h) + conv1x1(relu x)) + b2 + b0 (reference
    src/models/big_resnet.py:221-242)"""
    from studiogan_amd import functional as F, _lib as L
    N, Hl, Wl, C, Cout, C2, relu, bj = case
    monkeypatch.setenv("SG_CONV_Q_BJ", bj[:3])
    monkeypatch.setenv("SG_CONV_Q_DB", "1" if bj.endswith("db") else "0")
    dt = torch.bfloat16
    h = rnd((N, 2 * Hl, 2 * Wl, C), dt, 341)
    x = rnd((N, 2 * Hl, 2 * Wl, C2), dt, 342)
    w9 = rnd((Cout, 3, 3, C), dt, 343, 0.1)
    w0 = rnd((Cout, C2), dt, 344, 0.2)
    b2, b0 = rnd((Cout,), torch.float32, 345), rnd((Cout,), torch.float32, 346)
    ref = Q.pool_conv_torch(h.float(), w9.float(), relu) + b2 + b0
    img = C2 == 8                                    # ReLU on the main input only; filter image = mode 5 ([Cout][4 views][8] = w0 / 4 four times)
    xx = torch.relu(x.float()) if (relu and not img) else x.float()
    sk = torch.einsum("nhwc,oc->nhwo", xx, w0.float())
    ref = ref + 0.25 * (sk[:, 0::2, 0::2] + sk[:, 0::2, 1::2] + sk[:, 1::2, 0::2] + sk[:, 1::2, 1::2])
    wq = torch.empty(Cout, 16, C, dtype=dt, device="cuda:0")
    w9d, w0d = _dev(w9), _dev(w0)
    F.quad_pack_raw(w9d.data_ptr(), wq, 0, Cout, C)
    w0q = (w0d.float() * 0.25).to(dt)
    if img:
        w0q = w0q.repeat(1, 4).contiguous()
    y = F.conv2d_q_raw(_dev(h), wq.data_ptr(), L.Q_POOL, C, Cout, L.PIX_RELU if relu else 0, 0, bias=_dev(b2),
                       x2=_dev(x), w2q_ptr=w0q.data_ptr(), bias2=_dev(b0), x2_norelu=img)
    assert y is not None
    torch.cuda.synchronize()
    check(f"conv_q fused skip {case}", y.float().cpu(), ref, 6e-3)


def test_quad_pack_batch_matches_reference(sg):
    """every image of a network in one launch (what the weight bank runs behind sg_sn_forward): modes 0-3 and the scaled skip filter (mode 4)"""
    from studiogan_amd import _lib as L
    dt = torch.bfloat16
    specs = [(0, 96, 64), (1, 64, 96), (2, 32, 192), (3, 96, 32), (4, 192, 96), (0, 64, 32), (5, 96, 8)]
    srcs, dsts, refs = [], [], []
    arr = (L.QuadItem * len(specs))()
    for j, (mode, M, Cs) in enumerate(specs):
        if mode == 5:
            w = rnd((M, Cs), dt, 400 + j, 0.3)
            refs.append((w.double() * 0.25).repeat(1, 4))
            dst = torch.empty(M, 4 * Cs, dtype=dt, device="cuda:0")
        elif mode == 4:
            w = rnd((M, Cs), dt, 400 + j, 0.3)
            refs.append(w.double() * 0.25)
            dst = torch.empty(M, Cs, dtype=dt, device="cuda:0")
        else:
            w = rnd((M, 3, 3, Cs), dt, 400 + j, 0.1)
            refs.append(Q.quad_pack_ref(w.double(), mode).reshape(M, 16, Cs))
            dst = torch.empty(M, 16, Cs, dtype=dt, device="cuda:0")
        srcs.append(_dev(w)); dsts.append(dst)
        arr[j].src, arr[j].dst, arr[j].M, arr[j].Cs, arr[j].mode = srcs[-1].data_ptr(), dst.data_ptr(), M, Cs, mode
    tab = torch.frombuffer(bytearray(arr), dtype=torch.uint8).to("cuda:0")
    L.call("sg_quad_pack_batch", L.BF16, tab.data_ptr(), arr, len(specs), L.stream())
    torch.cuda.synchronize()
    for j, (mode, M, Cs) in enumerate(specs):
        check(f"quad pack batch item {j} mode {mode}", dsts[j].float().cpu(), refs[j], 4e-3)


@pytest.mark.parametrize("kind", ["quad_up", "quad_up_128", "quad_pool", "v4_skip"])
def test_conv_epilogue_bn_statistics(sg, kind, monkeypatch):
    """Batch-norm statistics taken in the producing convolution's epilogue (conv_v2.h sg_conv_epilogue `stats`) + sg_bn_stats_from_tiles
    against the sums of the stored bf16 result (what the separate statistics pass, csrc/norm.hip k_bn_partial_stream, reads back)."""
    from studiogan_amd import functional as F, _lib as L
    dt = torch.bfloat16
    N, Hl, Wl, C, Cout = 3, 8, 16, 64, 96          # J = 384 low-resolution pixels: a full and a partial 256-pixel tile
    w9 = rnd((Cout, 3, 3, C), dt, 501, 0.1)
    bias = rnd((Cout,), torch.float32, 502)
    w9d = _dev(w9)
    F._STATS_OFFER[0] = None
    if kind.startswith("quad"):
        form = L.Q_UP if "up" in kind else L.Q_POOL
        monkeypatch.setenv("SG_CONV_Q_BJ", "128" if kind.endswith("128") else "256")
        x = rnd((N, Hl, Wl, C) if form == L.Q_UP else (N, 2 * Hl, 2 * Wl, C), dt, 503)
        wq = torch.empty(Cout, 16, C, dtype=dt, device="cuda:0")
        F.quad_pack_raw(w9d.data_ptr(), wq, form, Cout, C)
        y = F.conv2d_q_raw(_dev(x), wq.data_ptr(), form, C, Cout, 0, 0, bias=_dev(bias), stats=True)
    else:
        x = rnd((N, 2 * Hl, 2 * Wl, C), dt, 503)
        x2 = rnd((N, Hl, Wl, 32), dt, 504)
        w0 = rnd((Cout, 32), dt, 505, 0.2)
        w0d = _dev(w0)
        y = F.conv2d_skip_raw(_dev(x), w9d.data_ptr(), C, Cout, _dev(x2), w0d.data_ptr(), 32, True, 0, 0, bias=_dev(bias), bias2=_dev(bias), stats=True)
    assert y is not None and F._STATS_OFFER[0] is not None, "no statistics were offered"
    _, ptr, shape, st, rows, Cc, ver = F._STATS_OFFER[0]
    assert ptr == y.data_ptr() and Cc == Cout and ver == y._version
    partial = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda:0")
    L.call("sg_bn_stats_from_tiles", st.data_ptr(), rows, Cout, partial.data_ptr(), L.stream())
    torch.cuda.synchronize()
    yd = y.double().cpu().reshape(-1, Cout)
    got = partial.cpu().reshape(Cout, 2)
    check(f"epilogue BN statistics {kind}: sum", got[:, 0], yd.sum(0), 1e-5)
    check(f"epilogue BN statistics {kind}: sum of squares", got[:, 1], (yd * yd).sum(0), 1e-5)
    # the offer is good for the untouched tensor only: an in-place write between the convolution and its batch norm withdraws it
    F._SEQ[0] = F._STATS_OFFER[0][0] + 1          # (the batch norm would be the very next operator)
    y.add_(1.0)
    assert F._take_stats(y) is None, "a stale statistics offer was taken after an in-place write to the tensor"
