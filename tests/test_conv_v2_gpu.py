"""GPU: the second-generation convolution kernel (csrc/conv_v2.h: LDS-DMA staging, 8 waves, BK=64) against CPU fp64 on
shapes that select each of its tile configurations, with every fused flag. SG_CONV_V2=force bypasses the
"enough tiles to fill 256 CUs" heuristic so that test-sized problems take the v2 path; the same shapes are also run
with SG_CONV_V2=0 (first-generation kernel) and the two results must agree to bf16 rounding."""
import os

import pytest
import torch
import torch.nn.functional as TF

from util import check
from test_kernels_gpu import rnd, nhwc, nchw, _conv_ref

pytestmark = pytest.mark.gpu

CASES = [
    # N, Cin, Cout, H, R, relu, up, pool
    (2, 64, 96, 16, 3, False, False, False),
    (2, 96, 96, 16, 3, True, False, True),
    (2, 96, 192, 16, 3, False, True, False),
    (1, 128, 128, 16, 3, True, True, True),
    (2, 192, 384, 16, 1, False, False, False),
    (3, 72, 96, 12, 3, False, False, False),      # cpt = 9: k-tiles straddle taps; J not a multiple of 256
    (2, 96, 288, 16, 3, True, False, False),
    (2, 24, 96, 16, 3, False, False, False),      # thin / odd chunk counts (cpt = 3, 5, 6): position recomputed from the chunk index
    (2, 40, 128, 16, 3, True, False, False),
    (2, 48, 96, 16, 3, True, True, False),
    (2, 64, 64, 16, 3, False, False, False),      # cout count no tile divides (64 on the 96-wide tile: weight rows >= 64 zero-filled, tail chunks not stored)
    (3, 96, 160, 17, 1, False, False, False),     # InceptionV3-like: 17 x 17 images (not a power of two), 160 couts on the 192-wide tile
    (2, 64, 320, 8, 3, True, False, False),       # 320 couts: two tiles of 192, the second one 2/3 full
]


@pytest.mark.parametrize("case", CASES)
def test_conv_v2_matches_reference_and_v1(sg, case):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, R, relu, up, pool = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    pad = R // 2
    x = rnd((N, Cin, H, H), dt, 71)
    w = rnd((Cout, Cin, R, R), dt, 72, 0.1)
    bias = rnd((Cout,), torch.float32, 73)
    Ho = H * (2 if up else 1)
    Hy = Ho // 2 if pool else Ho
    res = rnd((N, Cout, Hy, Hy), dt, 74)
    yref = _conv_ref(x, w, 1, pad, relu, up, pool, bias, res)
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    pf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    ef = L.EPI_POOL if pool else 0
    outs = {}
    for mode in ("force", "0"):
        os.environ["SG_CONV_V2"] = mode
        y = F.conv2d_raw(xd, wd.data_ptr(), Cin, Cout, R, R, 1, pad, pad, pf, ef, bias=bias.to(d), res=nhwc(res).to(d),
                         alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        outs[mode] = y.float().cpu()
    os.environ.pop("SG_CONV_V2", None)
    check(f"conv v2 {case}", nchw(outs["force"]), yref, 4e-3)
    check(f"conv v1 {case}", nchw(outs["0"]), yref, 4e-3)
    check(f"conv v2 vs v1 {case}", outs["force"], outs["0"], 4e-3)
    # data gradient through the same kernel (flipped weights, pooled-gradient broadcast / pooling-sum / ReLU mask)
    xr, wr = x.double().requires_grad_(True), w.double()
    y2 = _conv_ref(xr, wr, 1, pad, relu, up, pool, None, None)
    gy = rnd(tuple(y2.shape), dt, 75)
    y2.backward(gy.double())
    wdg = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(d)
    if Cin % 96 == 0 or Cin % 128 == 0:
        os.environ["SG_CONV_V2"] = "force"
        dx = F.conv2d_raw(nhwc(gy).to(d), wdg.data_ptr(), Cout, Cin, R, R, 1, R - 1 - pad, R - 1 - pad, L.PIX_UPSAMPLE if pool else 0,
                          L.EPI_POOL if up else 0, mask=xd if relu else None, alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        os.environ.pop("SG_CONV_V2", None)
        check(f"conv v2 dgrad {case}", nchw(dx.float().cpu()), xr.grad, 4e-3)


V3_CASES = [
    # N, Cin, Cout, H(in), relu, up, pool      -- 3x3, pad 1 (csrc/conv_v3.h: halo kernel)
    (2, 64, 192, 16, False, False, False),      # raster rows, one slice
    (2, 192, 192, 16, True, False, True),       # quad rows (pooling epilogue), three slices, ReLU on load
    (2, 96, 96, 16, False, False, False),       # C = 96: second slice half empty
    (1, 128, 128, 16, True, True, False),       # nearest x2 upsample on load
    (2, 128, 96, 8, False, True, True),         # upsample + pooling
    (8, 64, 96, 8, False, False, False),        # 8x8 images: a tile spans four images
    (5, 64, 96, 8, True, False, False),         # J = 320: partial last tile
    (3, 72, 96, 16, False, False, False),       # C = 72
    (5, 64, 128, 16, True, False, False),       # 128-wide cout tile
    (8, 96, 96, 128, True, False, True),        # the 512-pixel tile configuration (J = 131072), as D's first block
    (2, 384, 192, 8, True, False, False),       # six slices through the three-weight-buffer loop (cross-tap fragment prefetch)
    (2, 160, 128, 16, False, False, True),      # 128-wide tile, last slice half empty (nks = 2), pooling
    (1, 192, 384, 16, False, True, False),      # two cout tiles, upsample on load, three weight buffers
    (16, 64, 96, 4, True, False, False),        # 4x4 images (D's last blocks): a 256-pixel tile spans 16 images
    (8, 128, 96, 4, False, True, False),        # 4x4 source, upsampled to 8x8
    (2, 96, 8, 32, False, False, False),        # narrow cout tile (G's RGB layer: 3 couts padded to 8), cout-tail guard of the epilogue
]


@pytest.mark.parametrize("nw4", ["0", "1"])
@pytest.mark.parametrize("case", V3_CASES)
def test_conv_v3_matches_reference_and_v2(sg, case, nw4, monkeypatch):
    """nw4 = 1: the four-wave (one wave per SIMD) instantiations of the 192 / 128-wide tiles (SG_V3_NW4); other tiles are unaffected."""
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, relu, up, pool = case
    if nw4 == "1" and not (Cin % 64 == 0 and (Cout % 192 == 0 or Cout % 128 == 0)):
        pytest.skip("no four-wave variant of this tile")
    monkeypatch.setenv("SG_V3_NW4", nw4)
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    x = rnd((N, Cin, H, H), dt, 91)
    w = rnd((Cout, Cin, 3, 3), dt, 92, 0.1)
    bias = rnd((Cout,), torch.float32, 93)
    Ho = H * (2 if up else 1)
    Hy = Ho // 2 if pool else Ho
    res = rnd((N, Cout, Hy, Hy), dt, 94)
    big = N * Ho * Ho > 65536
    sel = [0, N - 1] if big else list(range(N))      # big case: CPU fp64 reference for the first and the last image only
    yref = _conv_ref(x[sel], w, 1, 1, relu, up, pool, bias, res[sel])
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    pf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    ef = L.EPI_POOL if pool else 0
    outs = {}
    os.environ["SG_CONV_V4"] = "0"                     # (conv_v4.h would take some of these shapes first)
    for name, v3, v2 in (("v3", "force", "force"), ("v2", "0", "force")):
        os.environ["SG_CONV_V3"], os.environ["SG_CONV_V2"] = v3, v2
        y = F.conv2d_raw(xd, wd.data_ptr(), Cin, Cout, 3, 3, 1, 1, 1, pf, ef, bias=bias.to(d), res=nhwc(res).to(d), alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        outs[name] = y.float().cpu()
    check(f"conv v3 {case}", nchw(outs["v3"])[sel], yref, 4e-3)
    check(f"conv v3 vs v2 {case}", outs["v3"], outs["v2"], 4e-3)
    # data gradient through the same kernel (flipped weights; ReLU mask / pooled-gradient broadcast / pooling-sum epilogues)
    if Cin % 96 == 0 or Cin % 128 == 0:
        xr, wr = x[sel].double().requires_grad_(True), w.double()
        y2 = _conv_ref(xr, wr, 1, 1, relu, up, pool, None, None)
        gy = rnd((N,) + tuple(y2.shape[1:]), dt, 95)
        y2.backward(gy[sel].double())
        wdg = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(d)
        os.environ["SG_CONV_V3"], os.environ["SG_CONV_V2"] = "force", "force"
        dx = F.conv2d_raw(nhwc(gy).to(d), wdg.data_ptr(), Cout, Cin, 3, 3, 1, 1, 1, L.PIX_UPSAMPLE if pool else 0,
                          L.EPI_POOL if up else 0, mask=xd if relu else None, alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        check(f"conv v3 dgrad {case}", nchw(dx.float().cpu())[sel], xr.grad, 4e-3)
    os.environ.pop("SG_CONV_V3", None)
    os.environ.pop("SG_CONV_V2", None)
    os.environ.pop("SG_CONV_V4", None)


V4_CASES = [
    # N, Cin, Cout, H(in), relu, up, pool      -- 3x3, pad 1 (csrc/conv_v4.h: small-workgroup halo kernel, 32-channel slices)
    (2, 96, 96, 16, False, False, False),       # raster rows, three slices, one cout tile
    (2, 96, 192, 16, True, False, True),        # quad rows (pooling epilogue), ReLU on load, two cout tiles
    (1, 192, 96, 16, True, True, False),        # nearest x2 upsample on load (G block conv1), six slices
    (2, 64, 128, 8, False, True, True),         # NB = 2 (64-wide cout tiles), upsample + pooling
    (8, 32, 64, 8, False, False, False),        # 8x8 images: a tile spans four images; one slice
    (5, 64, 96, 8, True, False, False),         # J = 320: partial last tile
    (4, 96, 96, 128, True, False, True),        # W = 128: the tile is one pair of image rows (D's first block, pooled)
    (4, 96, 96, 128, False, False, False),      # W = 128 raster
    (2, 192, 96, 64, True, True, False),        # 192 -> 96 up @128^2 (G's last block)
    (16, 64, 96, 4, True, False, False),        # 4x4 images: a 256-pixel tile spans 16 images
    (2, 384, 192, 8, True, False, False),       # twelve slices (a deep layer through the same loop)
]


@pytest.mark.parametrize("case", V4_CASES)
def test_conv_v4_matches_reference_and_v3(sg, case, monkeypatch):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, relu, up, pool = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    x = rnd((N, Cin, H, H), dt, 191)
    w = rnd((Cout, Cin, 3, 3), dt, 192, 0.1)
    bias = rnd((Cout,), torch.float32, 193)
    Ho = H * (2 if up else 1)
    Hy = Ho // 2 if pool else Ho
    res = rnd((N, Cout, Hy, Hy), dt, 194)
    big = N * Ho * Ho > 32768
    sel = [0, N - 1] if big else list(range(N))      # big case: CPU fp64 reference for the first and the last image only
    yref = _conv_ref(x[sel], w, 1, 1, relu, up, pool, bias, res[sel])
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    pf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    ef = L.EPI_POOL if pool else 0
    outs = {}
    os.environ["SG_CONV_V3"], os.environ["SG_CONV_V2"] = "force", "force"
    for name, v4 in (("v4", "all"), ("v3", "0")):
        os.environ["SG_CONV_V4"] = v4
        y = F.conv2d_raw(xd, wd.data_ptr(), Cin, Cout, 3, 3, 1, 1, 1, pf, ef, bias=bias.to(d), res=nhwc(res).to(d), alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        outs[name] = y.float().cpu()
    check(f"conv v4 {case}", nchw(outs["v4"])[sel], yref, 4e-3)
    check(f"conv v4 vs v3 {case}", outs["v4"], outs["v3"], 6e-3)       # two bf16-rounded results, each within 4e-3 of fp64
    # data gradient through the same kernel (flipped weights; ReLU mask / pooled-gradient broadcast / pooling-sum epilogues)
    if Cout % 32 == 0 and (Cin % 96 == 0 or Cin % 64 == 0):
        xr, wr = x[sel].double().requires_grad_(True), w.double()
        y2 = _conv_ref(xr, wr, 1, 1, relu, up, pool, None, None)
        gy = rnd((N,) + tuple(y2.shape[1:]), dt, 195)
        y2.backward(gy[sel].double())
        wdg = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(d)
        os.environ["SG_CONV_V4"] = "all"
        dx = F.conv2d_raw(nhwc(gy).to(d), wdg.data_ptr(), Cout, Cin, 3, 3, 1, 1, 1, L.PIX_UPSAMPLE if pool else 0,
                          L.EPI_POOL if up else 0, mask=xd if relu else None, alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        check(f"conv v4 dgrad {case}", nchw(dx.float().cpu())[sel], xr.grad, 4e-3)
    os.environ.pop("SG_CONV_V3", None)
    os.environ.pop("SG_CONV_V2", None)
    os.environ.pop("SG_CONV_V4", None)


SK_CASES = [
    # N, Cin, Cout, H(in), R, relu, up, pool      -- csrc/conv_sk.h: 1x1 with <= 192 channels and the 3x3 stem over 8 padded channels
    (2, 8, 96, 32, 3, False, False, False),      # the RGB stem (chunk = tap), halo zeros
    (3, 8, 96, 16, 3, False, False, True),       # stem-shaped with the pooling epilogue (quad rows); J = 768: partial wave strides
    (2, 8, 96, 16, 1, False, False, False),      # D's first skip: K = 8
    (2, 96, 16, 16, 1, False, False, False),     # attention theta / phi (12 couts padded to 16): cout tile 32 with a tail
    (2, 96, 48, 16, 1, True, False, False),      # attention g; ReLU on load
    (2, 48, 96, 16, 1, False, False, False),     # attention o: K = 48
    (2, 96, 192, 16, 1, False, False, True),     # D skip: 1x1 + pooling, two passes of 96 couts
    (2, 192, 96, 8, 1, False, True, False),      # G skip: nearest x2 upsample on load, K = 192
    (2, 192, 384, 16, 1, True, False, False),    # two cout tiles of 192 (three passes of 64), K = 192
    (5, 24, 192, 16, 1, False, False, False),    # K = 24 (zero-padded k tail), J = 1280
    (1, 64, 128, 16, 1, False, True, True),      # 128 couts (two passes of 64), upsample + pooling
]


@pytest.mark.parametrize("case", SK_CASES)
def test_conv_sk_matches_reference_and_v2(sg, case):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, R, relu, up, pool = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    pad = R // 2
    x = rnd((N, Cin, H, H), dt, 61)
    w = rnd((Cout, Cin, R, R), dt, 62, 0.1)
    bias = rnd((Cout,), torch.float32, 63)
    Ho = H * (2 if up else 1)
    Hy = Ho // 2 if pool else Ho
    res = rnd((N, Cout, Hy, Hy), dt, 64)
    msk = rnd((N, Cout, Hy, Hy), dt, 65)
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    pf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    ef = L.EPI_POOL if pool else 0
    al = 0.25 if pool else 1.0
    for what in ("res", "mask", "plain"):
        kw = dict(bias=bias.to(d)) if what != "mask" else {}
        if what == "res":
            kw["res"] = nhwc(res).to(d)
        if what == "mask":
            kw["mask"] = nhwc(msk).to(d)
        outs = {}
        for name, sk in (("sk", "force"), ("old", "0")):
            os.environ["SG_CONV_SK"] = sk
            y = F.conv2d_raw(xd, wd.data_ptr(), Cin, Cout, R, R, 1, pad, pad, pf, ef, alpha=al, **kw)
            torch.cuda.synchronize()
            outs[name] = y.float().cpu()
        os.environ.pop("SG_CONV_SK", None)
        yref = _conv_ref(x, w, 1, pad, relu, up, pool, bias if what != "mask" else None, res if what == "res" else None)
        if what == "mask":
            yref = yref * (msk.double() > 0)
        check(f"conv sk {what} {case}", nchw(outs["sk"]), yref, 4e-3)
        check(f"conv sk vs tile kernels {what} {case}", outs["sk"], outs["old"], 4e-3)


def test_conv_sk_full_size_stem_and_skip(sg):
    """BigGAN-128 D's stem (8 -> 96, 3x3 @128^2) and first skip (1x1 + pool) at batch 32: every wave runs several row blocks
    (software pipeline, out-of-range tail fetches); compared with the tile kernels on the whole tensor."""
    from studiogan_amd import functional as F, _lib as L
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    for (N, Cin, Cout, H, R, pool) in ((32, 8, 96, 128, 3, False), (32, 96, 192, 64, 1, True), (7, 96, 48, 64, 1, False)):
        x = rnd((N, H, H, Cin), dt, 51).to(d)
        w = rnd((Cout, R, R, Cin), dt, 52, 0.1).to(d)
        bias = rnd((Cout,), torch.float32, 53).to(d)
        outs = {}
        for name, sk in (("sk", "force"), ("old", "0")):
            os.environ["SG_CONV_SK"] = sk
            y = F.conv2d_raw(x, w.data_ptr(), Cin, Cout, R, R, 1, R // 2, R // 2, 0, L.EPI_POOL if pool else 0, bias=bias, alpha=0.25 if pool else 1.0)
            torch.cuda.synchronize()
            outs[name] = y.float().cpu()
        os.environ.pop("SG_CONV_SK", None)
        check(f"conv sk full size {(N, Cin, Cout, H, R, pool)}", outs["sk"], outs["old"], 4e-3)


WG_CASES = [
    # N, Cin, Cout, H, R, relu, up, pool
    (2, 64, 96, 16, 3, False, False, False),
    (2, 96, 96, 16, 3, True, False, True),
    (2, 96, 192, 16, 3, False, True, False),
    (1, 128, 128, 16, 3, True, True, True),
    (2, 192, 384, 16, 1, False, False, False),
    (3, 72, 200, 8, 3, True, False, False),       # partial I tile (648 rows), partial J tile, k tail
    (2, 64, 256, 16, 3, True, False, False),      # 256-wide cout tile (64 x 128 per wave), ReLU on load
    (2, 192, 512, 8, 1, False, False, False),     # two 256-wide cout tiles, 1x1
    (2, 96, 256, 16, 3, False, True, True),       # 256-wide tile with upsample-on-load and the pooled-gradient broadcast
]


@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad_v2_matches_reference_and_v1(sg, case):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, R, relu, up, pool = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    pad = R // 2
    x = rnd((N, Cin, H, H), dt, 81)
    w = rnd((Cout, Cin, R, R), dt, 82, 0.1)
    xr, wr = x.double(), w.double().requires_grad_(True)
    y = _conv_ref(xr, wr, 1, pad, relu, up, pool, None, None)
    gy = rnd(tuple(y.shape), dt, 83)
    y.backward(gy.double())
    Ho = H * (2 if up else 1)
    xd, gyd = nhwc(x).to(d), nhwc(gy).to(d)
    xf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    gf = L.PIX_UPSAMPLE if pool else 0
    outs = {}
    os.environ["SG_WGRAD_BJ256"] = "force"        # the 256-wide cout tile whenever Cout % 256 == 0 (test-sized problems are below its heuristic)
    for mode in ("force", "0"):
        os.environ["SG_CONV_V2"] = mode
        for splits in (0, 3):
            dw = torch.zeros((Cout, R, R, Cin), dtype=torch.float32, device=d)
            F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, R, Ho, Ho, 1, pad, pad, xf, gf, alpha=0.25 if pool else 1.0, splits=splits)
            torch.cuda.synchronize()
            outs[(mode, splits)] = dw.cpu()
            check(f"wgrad {mode} splits={splits} {case}", dw.cpu().permute(0, 3, 1, 2), wr.grad, 2e-3)
    os.environ.pop("SG_CONV_V2", None)
    os.environ.pop("SG_WGRAD_BJ256", None)
    check(f"wgrad v2 vs v1 {case}", outs[("force", 0)], outs[("0", 0)], 1e-4)


WSK_CASES = [
    # N, Cin, Cout, H(in), R, relu, up, pool      -- csrc/wgrad_sk.h: streaming weight gradient of the thin layers
    (2, 8, 96, 32, 3, False, False, False),      # D's RGB stem: nine taps out of one 3-row patch, halo zeros (W = 32: chunk = image row)
    (3, 8, 64, 64, 3, False, False, False),      # stem with two column blocks, W = 64 (two chunks per row), odd image count
    (2, 96, 8, 32, 3, False, False, False),      # G's RGB layer: roles of x and dy exchanged, tap shift negated
    (3, 64, 8, 64, 3, False, False, False),      # ... 64 input channels (ResNet generator), W = 64
    (2, 8, 96, 16, 1, False, False, False),      # D's first skip: 8 -> 96 (one row block, 24 of 32 rows empty)
    (2, 96, 16, 16, 1, False, False, False),     # attention theta / phi (12 couts padded to 16)
    (2, 96, 48, 16, 1, True, False, False),      # attention g, ReLU on the A operand
    (2, 48, 96, 16, 1, False, False, False),     # attention o
    (2, 96, 96, 16, 1, True, False, True),       # D skip shape: ReLU + pooled-gradient broadcast (dy at half resolution)
    (2, 192, 96, 8, 1, False, True, False),      # G skip: x at half resolution (nearest x2 on load), six row blocks
    (5, 192, 24, 16, 1, False, False, False),    # G attention theta: J = 24, 1280 pixels
    (2, 24, 96, 16, 1, False, False, False),     # 24 input channels: zero-filled channel tail of the row block
]


@pytest.mark.parametrize("case", WSK_CASES)
def test_wgrad_sk_matches_reference_and_tile_kernels(sg, case):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, R, relu, up, pool = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    pad = R // 2
    x = rnd((N, Cin, H, H), dt, 41)
    w = rnd((Cout, Cin, R, R), dt, 42, 0.1)
    xr, wr = x.double(), w.double().requires_grad_(True)
    y = _conv_ref(xr, wr, 1, pad, relu, up, pool, None, None)
    gy = rnd(tuple(y.shape), dt, 43)
    y.backward(gy.double())
    Ho = H * (2 if up else 1)
    xd, gyd = nhwc(x).to(d), nhwc(gy).to(d)
    xf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    gf = L.PIX_UPSAMPLE if pool else 0
    sig = torch.tensor([0.7], device=d)
    outs = {}
    for mode in ("1", "0"):
        os.environ["SG_WGRAD_SK"] = mode
        dw = torch.zeros((Cout, R, R, Cin), dtype=torch.float32, device=d)
        F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, R, Ho, Ho, 1, pad, pad, xf, gf, alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        outs[mode] = dw.cpu().clone()
        check(f"wgrad sk={mode} {case}", dw.cpu().permute(0, 3, 1, 2), wr.grad, 2e-3)
        # accumulation into dw (acml_steps > 1, shared weights) and the device-side scale (attention gate): dw += 0.7 * dW
        F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, R, Ho, Ho, 1, pad, pad, xf, gf, alpha=0.25 if pool else 1.0, alpha_ptr=sig)
        torch.cuda.synchronize()
        check(f"wgrad sk={mode} accumulate {case}", dw.cpu().permute(0, 3, 1, 2), 1.7 * wr.grad, 2e-3)
    os.environ.pop("SG_WGRAD_SK", None)
    check(f"wgrad sk vs tile kernels {case}", outs["1"], outs["0"], 1e-4)


def test_wgrad_sk_full_size_layers(sg):
    """The RGB layers and an attention 1x1 at the benchmark's resolution (batch 16): every wave runs many chunks (double-buffer
    hand-over, chunk stride = number of waves), compared with the tile kernels."""
    from studiogan_amd import functional as F, _lib as L
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    for (N, Cin, Cout, H, R) in ((16, 8, 96, 128, 3), (16, 96, 8, 128, 3), (16, 96, 48, 64, 1), (16, 192, 96, 64, 1)):
        x = rnd((N, H, H, Cin), dt, 31).to(d)
        gy = rnd((N, H, H, Cout), dt, 32).to(d)
        outs = {}
        for mode in ("1", "0"):
            os.environ["SG_WGRAD_SK"] = mode
            dw = torch.zeros((Cout, R, R, Cin), dtype=torch.float32, device=d)
            F.conv2d_wgrad_raw(x, gy, dw.data_ptr(), Cin, Cout, R, R, H, H, 1, R // 2, R // 2)
            torch.cuda.synchronize()
            outs[mode] = dw.cpu()
        os.environ.pop("SG_WGRAD_SK", None)
        check(f"wgrad sk full size {(N, Cin, Cout, H, R)}", outs["1"], outs["0"], 2e-3)


WV3_CASES = [
    # N, Cin, Cout, H(in), relu, up, pool      -- csrc/wgrad_v3.h: halo weight gradient of the wide-image 3x3 layers
    (2, 96, 96, 32, True, False, False),        # W = 32: chunk = two image rows, three cout blocks, ReLU on load
    (1, 32, 64, 64, False, False, False),       # W = 64: chunk = one row; one channel slice, two cout blocks (waves 2, 3 without extra tap)
    (1, 96, 192, 64, True, False, True),        # D block conv2: pooled gradient (dy at half resolution), two cout tiles
    (1, 192, 96, 32, True, True, False),        # G block conv1: ReLU + nearest x2 on load, six channel slices
    (1, 64, 128, 128, False, False, False),     # W = 128: two chunks per row (halo columns come from the neighbouring chunk)
    (3, 96, 96, 32, False, False, False),       # odd image count
    (1, 128, 192, 32, True, True, True),        # upsample on x and pooled dy at once (W = 64 after the upsample)
    (4, 64, 96, 16, True, False, False),        # W = 16: chunk = four image rows
    (4, 96, 64, 8, False, False, False),        # W = 8: chunk = a whole image, a k-step = two rows
    (4, 64, 64, 8, True, True, True),           # W = 16 after the upsample, pooled dy at 8 x 8
    (8, 96, 96, 4, True, False, False),         # W = 4 (round 4): chunk = four whole 4 x 4 images, each with its own 6 x 6 halo patch
    (4, 64, 192, 4, False, False, False),       # W = 4, one chunk, two cout tiles
]


@pytest.mark.parametrize("case", WV3_CASES)
def test_wgrad_v3_matches_reference_and_v2(sg, case):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, relu, up, pool = case
    R, pad = 3, 1
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    x = rnd((N, Cin, H, H), dt, 61)
    w = rnd((Cout, Cin, R, R), dt, 62, 0.1)
    xr, wr = x.double(), w.double().requires_grad_(True)
    y = _conv_ref(xr, wr, 1, pad, relu, up, pool, None, None)
    gy = rnd(tuple(y.shape), dt, 63)
    y.backward(gy.double())
    Ho = H * (2 if up else 1)
    xd, gyd = nhwc(x).to(d), nhwc(gy).to(d)
    xf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    gf = L.PIX_UPSAMPLE if pool else 0
    sig = torch.tensor([0.7], device=d)
    outs = {}
    for mode in ("force", "0"):
        os.environ["SG_WGRAD_V3"] = mode
        for splits in (0, 3):
            dw = torch.zeros((Cout, R, R, Cin), dtype=torch.float32, device=d)
            F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, R, Ho, Ho, 1, pad, pad, xf, gf, alpha=0.25 if pool else 1.0, splits=splits)
            torch.cuda.synchronize()
            outs[(mode, splits)] = dw.cpu().clone()
            check(f"wgrad v3={mode} splits={splits} {case}", dw.cpu().permute(0, 3, 1, 2), wr.grad, 2e-3)
        F.conv2d_wgrad_raw(xd, gyd, dw.data_ptr(), Cin, Cout, R, R, Ho, Ho, 1, pad, pad, xf, gf, alpha=0.25 if pool else 1.0, alpha_ptr=sig, splits=3)
        torch.cuda.synchronize()
        check(f"wgrad v3={mode} accumulate {case}", dw.cpu().permute(0, 3, 1, 2), 1.7 * wr.grad, 2e-3)
        # bias gradient riding along (column sums of the STORED dy -- the pooled one is read four times and scaled back), accumulating
        db = torch.full((Cout,), 0.5, dtype=torch.float32, device=d)
        dw2 = torch.zeros((Cout, R, R, Cin), dtype=torch.float32, device=d)
        fused = F.conv2d_wgrad_raw(xd, gyd, dw2.data_ptr(), Cin, Cout, R, R, Ho, Ho, 1, pad, pad, xf, gf, alpha=0.25 if pool else 1.0, dbias=db)
        torch.cuda.synchronize()
        assert fused == (mode == "force")
        if fused:
            check(f"wgrad v3 fused bias gradient {case}", db.cpu().double(), 0.5 + gy.double().sum((0, 2, 3)), 2e-3)
            check(f"wgrad v3 with bias: dw {case}", dw2.cpu(), outs[(mode, 0)], 1e-6)
        else:
            assert torch.equal(db.cpu(), torch.full((Cout,), 0.5))
    os.environ.pop("SG_WGRAD_V3", None)
    check(f"wgrad v3 vs v2 {case}", outs[("force", 0)], outs[("0", 0)], 1e-4)
    check(f"wgrad v3 split counts {case}", outs[("force", 0)], outs[("force", 3)], 1e-4)


def test_wgrad_v3_full_size_layers(sg):
    """The 96 / 192 / 384-channel layers at the benchmark's resolutions (batch 16): every workgroup walks many chunks (double-buffer
    hand-over, chunk stride = number of splits), XCD renumbering of the grid; compared with wgrad_v2."""
    from studiogan_amd import functional as F, _lib as L
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    for (N, Cin, Cout, H, xf, gf) in ((16, 96, 96, 128, L.PIX_RELU, 0), (16, 192, 192, 64, L.PIX_RELU, L.PIX_UPSAMPLE), (16, 384, 384, 32, 0, 0),
                                      (16, 192, 96, 128, L.PIX_RELU | L.PIX_UPSAMPLE, 0), (16, 64, 64, 32, L.PIX_RELU, 0),
                                      (64, 768, 768, 16, L.PIX_RELU, 0), (256, 768, 1536, 8, 0, 0)):
        hx = H // 2 if xf & L.PIX_UPSAMPLE else H
        hg = H // 2 if gf & L.PIX_UPSAMPLE else H
        x = rnd((N, hx, hx, Cin), dt, 71).to(d)
        gy = rnd((N, hg, hg, Cout), dt, 72).to(d)
        outs = {}
        for mode in ("1", "0"):
            os.environ["SG_WGRAD_V3"] = mode
            dw = torch.zeros((Cout, 3, 3, Cin), dtype=torch.float32, device=d)
            F.conv2d_wgrad_raw(x, gy, dw.data_ptr(), Cin, Cout, 3, 3, H, H, 1, 1, 1, xf, gf)
            torch.cuda.synchronize()
            outs[mode] = dw.cpu()
        os.environ.pop("SG_WGRAD_V3", None)
        check(f"wgrad v3 full size {(N, Cin, Cout, H, xf, gf)}", outs["1"], outs["0"], 2e-3)


@pytest.mark.parametrize("case", WV3_CASES)
def test_wgrad_v3_lean_matches_round4_kernel(sg, case, monkeypatch):
    """csrc/wgrad_v3l.h (the default since round 5: ReLU-on-load as a template parameter, bias gradient through v_dot2) against the round-4 kernel
    (SG_WGRAD_V3_LEAN=0, csrc/wgrad_v3.h): the same MFMAs in the same order -> dW bit for bit (one split layout), bias gradient to fp32 rounding."""
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, relu, up, pool = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    Ho = H * (2 if up else 1)
    hg = Ho // 2 if pool else Ho
    x = rnd((N, H, H, Cin), dt, 81).to(d)
    gy = rnd((N, hg, hg, Cout), dt, 82).to(d)
    xf = (L.PIX_RELU if relu else 0) | (L.PIX_UPSAMPLE if up else 0)
    gf = L.PIX_UPSAMPLE if pool else 0
    monkeypatch.setenv("SG_WGRAD_V3", "force")
    outs = {}
    for lean in ("0", "1"):
        monkeypatch.setenv("SG_WGRAD_V3_LEAN", lean)
        dw = torch.zeros((Cout, 3, 3, Cin), dtype=torch.float32, device=d)
        db = torch.zeros((Cout,), dtype=torch.float32, device=d)
        fused = F.conv2d_wgrad_raw(x, gy, dw.data_ptr(), Cin, Cout, 3, 3, Ho, Ho, 1, 1, 1, xf, gf, alpha=0.5, dbias=db)
        torch.cuda.synchronize()
        assert fused
        outs[lean] = (dw.cpu(), db.cpu())
    assert torch.equal(outs["0"][0], outs["1"][0]), case
    check(f"wgrad v3 lean bias gradient {case}", outs["1"][1], outs["0"][1], 1e-5)


SKIP_CASES = [
    # N, C (3x3 input), Cout, C2 (skip input), H (output res), relu, pool, up2      -- csrc/conv_v4.h SKIP: conv3x3(h) + conv1x1(up2?(x)) in one launch
    (2, 192, 192, 96, 16, True, True, False),     # DiscBlock 96 -> 192 (quad rows + pooling, ReLU on both inputs), W = 16: image-row parity swizzle
    (2, 384, 384, 192, 32, True, True, False),    # DiscBlock 192 -> 384 at the benchmarked resolution
    (3, 96, 96, 192, 32, False, False, True),     # GenBlock 192 -> 96: skip input at half resolution, nearest x2 on load
    (2, 192, 192, 384, 16, False, False, True),   # GenBlock 384 -> 192
    (2, 64, 128, 32, 16, True, False, False),     # 128-wide cout tile (NB = 2), no pooling, one skip slice
    (5, 96, 96, 96, 8, True, True, False),        # 8x8 images, J = 320: partial last tile (rows beyond the problem read the zero line)
    (1, 768, 768, 384, 16, True, True, False),    # a deep block (C > 384: only the fused path sends it to conv_v4)
]


@pytest.mark.parametrize("case", SKIP_CASES)
def test_conv_fused_skip_matches_reference_and_two_launches(sg, case):
    """sg_conv2d_fwd_skip (the residual block's 1x1 skip convolution as extra K-slices of its last 3x3 launch) against CPU fp64 and against
    the two chained launches it replaces; then functional.ConvSkipFn's backward (data + weight + bias gradients of both convolutions)."""
    from studiogan_amd import functional as F, _lib as L
    N, C, Cout, C2, H, relu, pool, up2 = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    H2 = H // 2 if up2 else H
    h = rnd((N, C, H, H), dt, 41)
    x = rnd((N, C2, H2, H2), dt, 42)
    w = rnd((Cout, C, 3, 3), dt, 43, 0.1)
    w0 = rnd((Cout, C2, 1, 1), dt, 44, 0.1)
    b, b0 = rnd((Cout,), torch.float32, 45), rnd((Cout,), torch.float32, 46)
    main = _conv_ref(h, w, 1, 1, relu, False, pool, b, None)
    skip = _conv_ref(x, w0, 1, 0, relu, up2, pool, b0, None)
    yref = main + skip
    hd, xd = nhwc(h).to(d), nhwc(x).to(d)
    wd, w0d = w.permute(0, 2, 3, 1).contiguous().to(d), w0.permute(0, 2, 3, 1).contiguous().to(d)
    pf = L.PIX_RELU if relu else 0
    ef = L.EPI_POOL if pool else 0
    al = 0.25 if pool else 1.0
    y = F.conv2d_skip_raw(hd, wd.data_ptr(), C, Cout, xd, w0d.data_ptr(), C2, up2, pf, ef, bias=b.to(d), bias2=b0.to(d), alpha=al)
    assert y is not None, "the fused kernel must take this shape"
    hh = F.conv2d_raw(hd, wd.data_ptr(), C, Cout, 3, 3, 1, 1, 1, pf, ef, bias=b.to(d), alpha=al)
    y2 = F.conv2d_raw(xd, w0d.data_ptr(), C2, Cout, 1, 1, 1, 0, 0, pf | (L.PIX_UPSAMPLE if up2 else 0), ef, bias=b0.to(d), res=hh, alpha=al)
    torch.cuda.synchronize()
    check(f"fused skip {case}", nchw(y.float().cpu()), yref, 4e-3)
    check(f"two launches {case}", nchw(y2.float().cpu()), yref, 6e-3)
    check(f"fused vs two launches {case}", y.float().cpu(), y2.float().cpu(), 1e-2)   # two bf16 results, the unfused one rounded twice


MASKRES_CASES = [
    # N, Cin, Cout, H, R, up, pool, engine env      -- relu-mask AND residual in one epilogue (the data gradient of a block's first convolution + GradLink)
    (2, 96, 96, 16, 3, False, False, {"SG_CONV_V4": "force"}),
    (2, 192, 96, 16, 3, False, True, {"SG_CONV_V4": "force"}),                          # pooling-sum epilogue (dgrad of an upsampling convolution)
    (2, 192, 192, 16, 3, False, False, {"SG_CONV_V4": "0", "SG_CONV_V3": "force"}),
    (2, 128, 128, 16, 3, True, False, {"SG_CONV_V4": "0", "SG_CONV_V3": "force"}),      # pooled-gradient broadcast on load
    (2, 192, 96, 16, 1, False, False, {"SG_CONV_SK": "0", "SG_CONV_V2": "force"}),
    (3, 72, 96, 12, 3, False, False, {"SG_CONV_V4": "0", "SG_CONV_V3": "0", "SG_CONV_V2": "0"}),   # generic engine
]


@pytest.mark.parametrize("case", MASKRES_CASES)
def test_conv_epilogue_mask_and_residual(sg, case, monkeypatch):
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, R, up, pool, env = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    pad = R // 2
    x = rnd((N, Cin, H, H), dt, 51)
    w = rnd((Cout, Cin, R, R), dt, 52, 0.1)
    Ho = H * (2 if up else 1)
    Hy = Ho // 2 if pool else Ho
    m = rnd((N, Cout, Hy, Hy), dt, 53)
    res = rnd((N, Cout, Hy, Hy), dt, 54)
    y0 = _conv_ref(x, w, 1, pad, False, up, pool, None, None)
    yref = y0 * (m.double() > 0) + res.double()
    pf = L.PIX_UPSAMPLE if up else 0
    ef = L.EPI_POOL if pool else 0
    wd = w.permute(0, 2, 3, 1).contiguous().to(d)
    y = F.conv2d_raw(nhwc(x).to(d), wd.data_ptr(), Cin, Cout, R, R, 1, pad, pad, pf, ef,
                     mask=nhwc(m).to(d), res=nhwc(res).to(d), alpha=0.25 if pool else 1.0)
    torch.cuda.synchronize()
    check(f"mask + residual {case[:7]}", nchw(y.float().cpu()), yref, 4e-3)


STRIDE2_CASES = [
    # N, Cin, Cout, H, R, pad      -- stride-2 convolutions on the conv_v2 tile kernels (InceptionV3's reduction layers)
    (4, 96, 96, 35, 3, 0),          # Mixed_6a.branch3x3dbl_3: 35 -> 17, valid
    (2, 288, 384, 35, 3, 0),        # Mixed_6a.branch3x3
    (3, 192, 320, 17, 3, 0),        # Mixed_7a.branch3x3_2: 17 -> 8, cout count no tile divides
    (2, 64, 128, 16, 3, 1),         # padded, even size (DCGAN-style 4x4 s2 has its own transposed path; this is the plain strided form)
    (2, 64, 96, 16, 1, 0),          # 1x1 stride 2
]


@pytest.mark.parametrize("case", STRIDE2_CASES)
def test_conv_v2_stride2_matches_reference_and_generic(sg, case, monkeypatch):
    from studiogan_amd import functional as F
    N, Cin, Cout, H, R, pad = case
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    x = rnd((N, Cin, H, H), dt, 81)
    w = rnd((Cout, Cin, R, R), dt, 82, 0.1)
    bias = rnd((Cout,), torch.float32, 83)
    yref = _conv_ref(x, w, 2, pad, False, False, False, bias, None)
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    outs = {}
    for mode in ("force", "0"):
        monkeypatch.setenv("SG_CONV_V2", mode)
        y = F.conv2d_raw(xd, wd.data_ptr(), Cin, Cout, R, R, 2, pad, pad, 0, 0, bias=bias.to(d))
        torch.cuda.synchronize()
        outs[mode] = y.float().cpu()
    assert tuple(nchw(outs["force"]).shape) == tuple(yref.shape)
    check(f"conv v2 stride 2 {case}", nchw(outs["force"]), yref, 4e-3)
    check(f"generic stride 2 {case}", nchw(outs["0"]), yref, 4e-3)
    check(f"conv v2 vs generic stride 2 {case}", outs["force"], outs["0"], 4e-3)


RS_CASES = [
    # N, Cin, Cout, H, relu_in, relu_out, bias, alpha, strip rows (SG_CONV_RS_SH; 0 = the launcher's choice)
    # -- csrc/conv_rs.h: the row-streaming kernel of the RGB layers (3x3, <= 32 output channels, 128-pixel-wide images, weights in registers)
    (2, 96, 8, 128, False, False, True, 1.0, 128),      # one strip per image: top / bottom image borders inside the walk (G's RGB layer at batch 256)
    (2, 96, 8, 128, False, False, False, 1.0, 0),       # the launcher's strips (8 rows at this batch): halo rows from the neighbouring strips
    (3, 96, 8, 128, True, True, True, 0.5, 32),         # ReLU on load, ReLU on store, scale
    (2, 64, 16, 128, False, False, True, 1.0, 16),      # 64 input channels (18 DMA pieces per row), two cout groups
    (2, 96, 32, 128, False, False, True, 1.0, 64),      # a full 32-cout tile
    (1, 96, 24, 64, False, False, True, 1.0, 4),        # 64 x 128 images (H != W), three cout groups, 4-row strips (the ring wraps inside the prologue)
]


@pytest.mark.parametrize("case", RS_CASES)
def test_conv_rs_matches_reference_and_halo_kernel(sg, case, monkeypatch):
    """The row-streaming kernel against CPU fp64 and against the halo kernel it replaces on these shapes (SG_CONV_RS=0), forward and -- as the
    data gradient of the discriminator's RGB stem runs it -- with the flipped [Cin][R][S][Cout] image."""
    from studiogan_amd import functional as F, _lib as L
    N, Cin, Cout, H, relu_in, relu_out, with_bias, alpha, sh = case
    Wd = 128
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    x = rnd((N, Cin, H, Wd), dt, 171)
    w = rnd((Cout, Cin, 3, 3), dt, 172, 0.1)
    bias = rnd((Cout,), torch.float32, 173) if with_bias else None
    yref = _conv_ref(x, w, 1, 1, relu_in, False, False, None, None) * alpha
    if bias is not None:
        yref = yref + bias.double().view(1, -1, 1, 1)
    if relu_out:
        yref = torch.relu(yref)
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    pf = L.PIX_RELU if relu_in else 0
    ef = L.EPI_RELU if relu_out else 0
    outs = {}
    for mode in ("force", "0"):
        monkeypatch.setenv("SG_CONV_RS", mode)
        if sh:
            monkeypatch.setenv("SG_CONV_RS_SH", str(sh))
        before = L.lib().sg_conv_rs_launches()
        y = F.conv2d_raw(xd, wd.data_ptr(), Cin, Cout, 3, 3, 1, 1, 1, pf, ef, bias=None if bias is None else bias.to(d), alpha=alpha)
        torch.cuda.synchronize()
        assert L.lib().sg_conv_rs_launches() - before == (1 if mode == "force" else 0), "the wrong engine took the problem"
        outs[mode] = y.float().cpu()
    check(f"conv rs {case}", nchw(outs["force"]), yref, 4e-3)
    check(f"conv rs vs halo kernel {case}", outs["force"], outs["0"], 4e-3)
    assert not torch.equal(outs["force"], torch.zeros_like(outs["force"]))


RS96_CASES = [
    # N, H, relu_in, pool, relu_out, bias, strip rows     -- csrc/conv_rs96.h: 96 -> 96 channels, 3x3, 128-pixel-wide images
    (2, 128, False, False, False, True, 128),       # G's last 3x3 (plain + bias), one strip per image
    (2, 128, True, True, False, True, 0),           # D's first-block tail: ReLU on load, 2x2 average pooling (alpha = 0.25), launcher's strips
    (3, 64, True, False, True, False, 16),          # 64 x 128 images, ReLU on store, no bias
    (2, 128, False, True, False, True, 2),          # pooling with two-row strips (every strip one pooled row)
]


@pytest.mark.parametrize("case", RS96_CASES)
def test_conv_rs96_matches_reference_and_halo_kernel(sg, case, monkeypatch):
    from studiogan_amd import functional as F, _lib as L
    N, H, relu_in, pool, relu_out, with_bias, sh = case
    Wd, Cc = 128, 96
    d = torch.device("cuda:0")
    dt = torch.bfloat16
    x = rnd((N, Cc, H, Wd), dt, 271)
    w = rnd((Cc, Cc, 3, 3), dt, 272, 0.05)
    bias = rnd((Cc,), torch.float32, 273) if with_bias else None
    yref = _conv_ref(x, w, 1, 1, relu_in, False, pool, bias, None)
    if relu_out:
        yref = torch.relu(yref)
    xd, wd = nhwc(x).to(d), w.permute(0, 2, 3, 1).contiguous().to(d)
    pf = L.PIX_RELU if relu_in else 0
    ef = (L.EPI_POOL if pool else 0) | (L.EPI_RELU if relu_out else 0)
    outs = {}
    for mode in ("force", "0"):
        monkeypatch.setenv("SG_CONV_RS96", mode)
        if sh:
            monkeypatch.setenv("SG_CONV_RS_SH", str(sh))
        before = L.lib().sg_conv_rs_launches()
        y = F.conv2d_raw(xd, wd.data_ptr(), Cc, Cc, 3, 3, 1, 1, 1, pf, ef, bias=None if bias is None else bias.to(d), alpha=0.25 if pool else 1.0)
        torch.cuda.synchronize()
        assert L.lib().sg_conv_rs_launches() - before == (1 if mode == "force" else 0), "the wrong engine took the problem"
        outs[mode] = y.float().cpu()
    check(f"conv rs96 {case}", nchw(outs["force"]), yref, 4e-3)
    check(f"conv rs96 vs halo kernel {case}", outs["force"], outs["0"], 4e-3)
