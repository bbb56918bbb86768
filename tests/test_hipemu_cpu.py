"""Lane-accurate CPU runs of the gfx950 kernel SOURCES (tests/hipemu: the csrc translation unit compiled for the host against an interpreter of MFMA /
ds_read_b64_tr_b16 / LDS-DMA / barriers; see tests/hipemu/include/hip/hip_runtime.h). Two uses:

  * pin the interpreter: kernels that have passed their GPU parity tests (tests/test_conv_v2_gpu.py) must reproduce the fp64 restatement of
    include/sgamd.h's formula here too, under every DMA-completion / wave-scheduling mode;
  * check kernels written without GPU time against their GPU-verified predecessors before their first launch (SG_WGRAD_V3_LEAN, ...).

Not a product path and not an oracle of the reference: it executes the product's own kernel code."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
import emu  # noqa: E402

pytestmark = pytest.mark.skipif(not emu.available(), reason="host clang++ of the ROCm toolchain not found")

MODES = [dict(dma_late=0, greedy=0, seed=0), dict(dma_late=1, greedy=1, seed=1), dict(dma_late=0, greedy=1, seed=2), dict(dma_late=1, greedy=1, seed=3)]
# (N, H, W, C, Cout): every chunk shape of wgrad_v3.h (WC = 4 / 8 / 16 / 32 / 64) with both cout tiles (NB = 2: 64, NB = 3: 96 / 192)
V3_SHAPES = [(4, 4, 4, 32, 64), (8, 4, 4, 64, 96), (2, 8, 8, 32, 64), (1, 8, 8, 64, 192), (1, 16, 16, 32, 96), (2, 32, 32, 32, 64), (1, 64, 64, 32, 64),
             (1, 64, 128, 32, 96)]


@pytest.fixture(scope="module")
def wg():
    return emu.load("conv_wgrad")


def _data(shape, seed):
    N, H, W, Cin, Cout = shape
    rng = np.random.default_rng(seed)
    return emu.to_bf16(rng.standard_normal((N, H, W, Cin)).astype(np.float32)), emu.to_bf16(rng.standard_normal((N, H, W, Cout)).astype(np.float32))


def test_desc_layout_matches_product():
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("sg_lib_fields", os.path.join(here, "..", "pytorch-studiogan_amd", "_lib.py"))
    src = open(spec.origin).read()
    # the product's field list, textually (importing _lib would build / load libsgamd.so)
    import re
    m = re.search(r"class ConvWgradDesc\(C.Structure\):\s+_fields_ = \[(.*?)\]\n", src, re.S)
    names = re.findall(r'\("(\w+)", (_\w+)\)', m.group(1))
    assert [(n, t) for n, t in names] == [(n, {emu._vp: "_vp", emu._i: "_i", emu._f: "_f", emu._ll: "_ll"}[t]) for n, t in emu.WGRAD_FIELDS]


@pytest.mark.parametrize("shape", V3_SHAPES)
def test_wgrad_v3_pins_interpreter(wg, shape):
    """the GPU-verified halo weight gradient, as shipped, through the interpreter: fp64 formula within fp32 accumulation noise"""
    x, dy = _data(shape, 11)
    for k, mode in enumerate(MODES if shape[0] * shape[1] * shape[2] <= 2048 else MODES[1:2]):
        emu.config(wg, **mode)
        c0 = emu.counters(wg)
        for xf in (0, emu.PIX_RELU):
            dw, db, sp = emu.conv_wgrad(wg, x, dy, shape[4], x_flags=xf, bias=True, env={"SG_WGRAD_V3": "f", "SG_WGRAD_V3_LEAN": "0"})
            ref, rb = emu.wgrad_ref(x, dy, x_flags=xf)
            assert np.abs(dw - ref).max() <= 2e-6 * np.abs(ref).max(), (shape, mode, xf)
            assert db is not None and np.abs(db - rb).max() <= 2e-6 * np.abs(rb).max()
        c1 = emu.counters(wg)
        assert c1["mfma"] > c0["mfma"] and c1["tr_reads"] > c0["tr_reads"] and c1["dma_ops"] > c0["dma_ops"]     # the halo kernel ran, not the GEMM path


@pytest.mark.parametrize("shape", V3_SHAPES)
def test_wgrad_v3_lean_equals_shipped(wg, shape):
    """wgrad_v3l.h (SG_WGRAD_V3_LEAN=1): same MFMAs in the same order -> dW bit for bit; bias gradient summed in another order -> fp32 rounding"""
    x, dy = _data(shape, 12)
    emu.config(wg, dma_late=1, greedy=1, seed=5)
    for xf in (0, emu.PIX_RELU):
        a, ab, _ = emu.conv_wgrad(wg, x, dy, shape[4], x_flags=xf, bias=True, alpha=0.5, env={"SG_WGRAD_V3": "f", "SG_WGRAD_V3_LEAN": "0"})
        b, bb, _ = emu.conv_wgrad(wg, x, dy, shape[4], x_flags=xf, bias=True, alpha=0.5, env={"SG_WGRAD_V3": "f", "SG_WGRAD_V3_LEAN": "1"})
        assert np.array_equal(a, b), (shape, xf)
        assert np.abs(ab - bb).max() <= 1e-6 * np.abs(ab).max()


def test_wgrad_v3_upsampled_operands(wg):
    """x read through the nearest-neighbour upsampling (generator conv2d1 without the quad form) and dy read as the pooled gradient (g_up)"""
    rng = np.random.default_rng(3)
    N, Hs, Ws, Cin, Cout = 2, 8, 8, 32, 64
    x = emu.to_bf16(rng.standard_normal((N, Hs, Ws, Cin)).astype(np.float32))
    dy_full = emu.to_bf16(rng.standard_normal((N, 2 * Hs, 2 * Ws, Cout)).astype(np.float32))
    dy_low = emu.to_bf16(rng.standard_normal((N, Hs, Ws, Cout)).astype(np.float32))
    x_full = emu.to_bf16(rng.standard_normal((N, 2 * Hs, 2 * Ws, Cin)).astype(np.float32))
    emu.config(wg, dma_late=1, greedy=1, seed=9)
    for lean in ("0", "1"):
        env = {"SG_WGRAD_V3": "f", "SG_WGRAD_V3_LEAN": lean}
        dw, db, _ = emu.conv_wgrad(wg, x, dy_full, Cout, x_flags=emu.PIX_UPSAMPLE, bias=True, env=env)
        ref, rb = emu.wgrad_ref(x, dy_full, x_flags=emu.PIX_UPSAMPLE)
        assert np.abs(dw - ref).max() <= 2e-6 * np.abs(ref).max() and np.abs(db - rb).max() <= 2e-6 * np.abs(rb).max()
        dw, db, _ = emu.conv_wgrad(wg, x_full, dy_low, Cout, x_flags=emu.PIX_RELU, g_flags=emu.PIX_UPSAMPLE, bias=True, env=env)
        ref, rb = emu.wgrad_ref(x_full, dy_low, x_flags=emu.PIX_RELU, g_flags=emu.PIX_UPSAMPLE)
        assert np.abs(dw - ref).max() <= 2e-6 * np.abs(ref).max()
        if db is not None:
            assert np.abs(db - rb).max() <= 2e-6 * np.abs(rb).max()


@pytest.mark.skipif(os.environ.get("SG_EMU_NET") != "1", reason="builds a second, deliberately broken copy of conv_wgrad.hip (~1 min): SG_EMU_NET=1 (last run: profiles/r04_hipemu_nets.txt)")
def test_interpreter_catches_a_missing_wait(wg, tmp_path):
    """the adversarial DMA mode is not decoration: the same kernel with its `s_waitcnt vmcnt(0)` removed must FAIL under late completion"""
    import translate
    src = open(os.path.join(translate.CSRC, "wgrad_v3.h")).read()
    assert 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");' in src
    broken = translate.translate(src.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', ""))
    assert "waitcnt_vm" not in broken
    # build a private copy of the translated tree with the broken header
    import shutil, subprocess
    d = tmp_path / "src"
    shutil.copytree(emu.SRC, d)
    (d / "wgrad_v3.h").write_text(broken)
    lib = tmp_path / "libbroken.so"
    subprocess.run([emu.CXX, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(emu.HERE, "include"), "-I" + str(d),
                    "-Wno-unknown-attributes", "-Wno-unused-value", "-ffp-contract=off", str(d / "conv_wgrad.hip"), os.path.join(emu.HERE, "stubs.cpp"),
                    os.path.join(emu.HERE, "rt.cpp"), "-o", str(lib)], check=True)
    import ctypes
    bl = ctypes.CDLL(str(lib))
    bl.sg_last_error.restype = ctypes.c_char_p
    x, dy = _data((2, 8, 8, 32, 64), 4)
    ref, _ = emu.wgrad_ref(x, dy)
    emu.config(bl, dma_late=0, greedy=0, seed=0)
    dw, _, _ = emu.conv_wgrad(bl, x, dy, 64, env={"SG_WGRAD_V3": "f"})
    assert np.abs(dw - ref).max() <= 2e-6 * np.abs(ref).max()          # eager completion hides the bug, like a lucky GPU run
    emu.config(bl, dma_late=1, greedy=1, seed=1)
    dw, _, _ = emu.conv_wgrad(bl, x, dy, 64, env={"SG_WGRAD_V3": "f"})
    assert not np.abs(dw - ref).max() <= 1e-3 * np.abs(ref).max()      # late completion exposes it


# ---- quad convolutions (conv_q.hip) -------------------------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# (N, Hl, Wl, C, Cout) on the LOW-resolution grid: every chunk shape of wgrad_q.h (WC = 4 / 8 / 16 / 32 / 64), S = 1 and 2 slices, NB = 2 and 3
Q_SHAPES = [(4, 4, 4, 32, 64), (4, 4, 4, 64, 192), (2, 8, 8, 64, 96), (1, 16, 16, 32, 96), (1, 32, 32, 64, 64), (1, 64, 64, 32, 64)]


@pytest.fixture(scope="module")
def cq():
    return emu.load("conv_q")


def _t64(a):
    import torch
    return torch.from_numpy(emu.from_bf16(a).astype(np.float64))


def _qdata(form, shape, seed):
    N, Hl, Wl, Cin, Cout = shape
    rng = np.random.default_rng(seed)
    xs = (N, 2 * Hl, 2 * Wl, Cin) if form == emu.Q_POOL else (N, Hl, Wl, Cin)
    gs = (N, Hl, Wl, Cout) if form == emu.Q_POOL else (N, 2 * Hl, 2 * Wl, Cout)
    return emu.to_bf16(rng.standard_normal(xs).astype(np.float32)), emu.to_bf16(rng.standard_normal(gs).astype(np.float32))


@pytest.mark.parametrize("form", [emu.Q_POOL, emu.Q_UP])
@pytest.mark.parametrize("shape", Q_SHAPES)
def test_conv_q_forward_pins_interpreter(cq, form, shape):
    """the GPU-verified quad forward kernel (LDS-DMA patch + four weight buffers with counted waits, swizzled ds_read_b128 fragments, the staged
    epilogue with bias / ReLU mask / residual) against tests/quad_ref.py on the kernel's own filter image; sg_quad_pack against its restatement"""
    import torch
    import quad_ref as Q
    N, Hl, Wl, Cin, Cout = shape
    rng = np.random.default_rng(21)
    x, _ = _qdata(form, shape, 21)
    w9 = emu.to_bf16((0.1 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
    emu.config(cq, dma_late=1, greedy=1, seed=6)
    wq = emu.quad_pack(cq, w9, form)
    wq_ref = Q.quad_pack_ref(_t64(w9), form)
    assert np.array_equal(wq, emu.to_bf16(wq_ref.numpy().astype(np.float32)))                  # fp32 sums of bf16 taps, rounded once
    oshape = (N, Hl, Wl, Cout) if form == emu.Q_POOL else (N, 2 * Hl, 2 * Wl, Cout)
    bias = rng.standard_normal(Cout).astype(np.float32)
    mask = emu.to_bf16(rng.standard_normal(oshape).astype(np.float32))
    res = emu.to_bf16(rng.standard_normal(oshape).astype(np.float32))
    for mode in (dict(dma_late=1, greedy=1, seed=6), dict(dma_late=0, greedy=1, seed=7)):
        emu.config(cq, **mode)
        for relu in (False, True):
            out = emu.conv_q(cq, form, x, wq, Cout, relu_in=relu, bias=bias, mask=mask, res=res, alpha=0.5)
            xx = _t64(x).clamp(min=0) if relu else _t64(x)
            ref = 0.5 * Q.convq_ref(xx, _t64(wq), form) + torch.from_numpy(bias).double()
            ref = ref * (_t64(mask) > 0) + _t64(res)
            got = _t64(out)
            assert (got - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-6, (form, shape, relu)     # one bf16 rounding of the result


@pytest.mark.parametrize("form", [emu.Q_POOL, emu.Q_UP])
@pytest.mark.parametrize("shape", Q_SHAPES)
def test_wgrad_q_pins_interpreter_and_lean_equals_shipped(cq, form, shape):
    """shipped sg_wgrad_q_kernel + k_quad_reduce_fold against the fp64 restatement; wgrad_ql.h (SG_WGRAD_Q_LEAN=1: DMA addresses once per workgroup,
    ReLU as a template parameter, bias gradient through v_dot2) bit for bit against the shipped kernel"""
    import torch
    import quad_ref as Q
    x, dy = _qdata(form, shape, 22)
    emu.config(cq, dma_late=1, greedy=1, seed=8)
    for relu in (False, True):
        a, ab, _ = emu.conv_q_wgrad(cq, form, x, dy, relu_in=relu, bias=True, alpha=0.5, env={"SG_WGRAD_Q": "f", "SG_WGRAD_Q_LEAN": "0"})
        xx = _t64(x).clamp(min=0) if relu else _t64(x)
        ref = 0.5 * Q.quad_fold_ref(Q.wgradq_ref(xx, _t64(dy), form), form)
        rb = _t64(dy).sum((0, 1, 2))
        assert (torch.from_numpy(a).double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item(), (form, shape, relu)
        assert (torch.from_numpy(ab).double() - rb).abs().max().item() <= 2e-6 * rb.abs().max().item()
        for lean in ("1",):              # the default kernel: lean loop + the two-deep register pipeline over the k-steps of a chunk (counted lgkmcnt waits)
            b, bb, _ = emu.conv_q_wgrad(cq, form, x, dy, relu_in=relu, bias=True, alpha=0.5, env={"SG_WGRAD_Q": "f", "SG_WGRAD_Q_LEAN": lean})
            assert np.array_equal(a, b), (form, shape, relu, lean)
            assert np.abs(ab - bb).max() <= 1e-6 * np.abs(ab).max()


@pytest.mark.parametrize("form", [emu.Q_POOL, emu.Q_UP])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 96), (1, 32, 32, 96, 192), (1, 16, 64, 32, 64), (3, 8, 8, 64, 96)])      # (the last: one ragged tile)
def test_conv_q_is_schedule_independent(cq, form, shape):
    """conv_q.h's single-buffered loop under late DMA completion and three seeded wave orders: the same bf16 output bit for bit (its counted waits and buffer
    re-use hold whatever the schedule); with the fused 1x1 skip of the POOL form too. (Round 4 compared three loop variants here -- weights three taps ahead,
    tap pairs, one-sided halo: measured in round 5, none faster, removed.)"""
    N, Hl, Wl, Cin, Cout = shape
    rng = np.random.default_rng(31)
    x, _ = _qdata(form, shape, 31)
    w9 = emu.to_bf16((0.1 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
    wq = emu.quad_pack(cq, w9, form)
    bias = rng.standard_normal(Cout).astype(np.float32)
    kw = {}
    if form == emu.Q_POOL:
        kw = dict(x2=emu.to_bf16(rng.standard_normal((N, 2 * Hl, 2 * Wl, 32)).astype(np.float32)), w2q=emu.to_bf16((0.1 * rng.standard_normal((Cout, 32))).astype(np.float32)),
                  bias2=rng.standard_normal(Cout).astype(np.float32))
    outs = {}
    for la in ("0",):
        for seed in (1, 2, 3):
            emu.config(cq, dma_late=1, greedy=1, seed=seed)
            c0 = emu.counters(cq)
            outs[(la, seed)] = emu.conv_q(cq, form, x, wq, Cout, relu_in=True, bias=bias, env={"SG_CONV_Q_BJ": "256", "SG_CONV_Q_DB": "0"}, **kw).copy()
            assert emu.counters(cq)["dma_ops"] > c0["dma_ops"]
    ref = outs[("0", 1)]
    for k, v in outs.items():
        assert np.array_equal(v, ref), k


@pytest.mark.parametrize("form", [emu.Q_POOL, emu.Q_UP])
@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 192), (5, 8, 8, 32, 96)])      # 3 and 2 pixel tiles (the second: ragged last tile), two / one cout tiles
def test_conv_q_tile_order_groups(cq, form, shape):
    """conv_q.h's weight-stationary tile order (gj > 1: groups of pixel tiles share one weight set; round 6, the deep layers' L2 misses) is a pure renumbering of the
    workgroups: bit-identical output for group sizes that divide the pixel tiles, that leave a ragged last group, and that exceed them"""
    N, Hl, Wl, Cin, Cout = shape
    rng = np.random.default_rng(41)
    x, _ = _qdata(form, shape, 41)
    w9 = emu.to_bf16((0.1 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
    wq = emu.quad_pack(cq, w9, form)
    bias = rng.standard_normal(Cout).astype(np.float32)
    emu.config(cq, dma_late=1, greedy=1, seed=5)
    ref = emu.conv_q(cq, form, x, wq, Cout, relu_in=True, bias=bias, env={"SG_CONV_Q_BJ": "256", "SG_CONV_Q_GJ": "1"}).copy()
    for gj in ("2", "3", "8"):
        out = emu.conv_q(cq, form, x, wq, Cout, relu_in=True, bias=bias, env={"SG_CONV_Q_BJ": "256", "SG_CONV_Q_GJ": gj})
        assert np.array_equal(out, ref), (form, shape, gj)


# ---- conv_v4.h (3x3 halo kernel of the <= 384-channel layers) -----------------------------------------------------------------------------------------------
V4_CASES = [
    # N, H, C, Cout, relu_in, up, pool
    (2, 16, 64, 96, True, False, False),        # NB = 3, two channel slices (ring of four buffers: slice 1 starts at buffer 1)
    (1, 32, 96, 64, False, False, False),       # NB = 2, three slices
    (2, 16, 64, 192, True, False, True),        # quad row order + pooling epilogue, two cout tiles
    (2, 8, 96, 96, True, True, False),          # nearest x2 upsampling on load
    (1, 16, 32, 96, False, False, False),       # ONE slice: the tail waits of the last slice from tap 0 on
]


@pytest.mark.parametrize("case", V4_CASES)
def test_conv_v4_pins_interpreter(cq, case):
    """the GPU-verified halo kernel against torch on the same bf16 inputs, bit-identical under late DMA completion and three wave orders"""
    import torch
    import torch.nn.functional as TF
    N, H, Cin, Cout, relu, up, pool = case
    rng = np.random.default_rng(41)
    x = emu.to_bf16(rng.standard_normal((N, H, H, Cin)).astype(np.float32))
    w = emu.to_bf16((0.1 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
    bias = rng.standard_normal(Cout).astype(np.float32)
    env = {"SG_CONV_V4": "force", "SG_CONV_V3": "0", "SG_QUAD": "0", "SG_CONV_RS": "0"}
    xr = _t64(x).permute(0, 3, 1, 2)
    if relu:
        xr = xr.clamp(min=0)
    if up:
        xr = TF.interpolate(xr, scale_factor=2, mode="nearest")
    ref = TF.conv2d(xr, _t64(w).permute(0, 3, 1, 2), None, padding=1)
    if pool:
        ref = TF.avg_pool2d(ref, 2)
    ref = ref + torch.from_numpy(bias).double()[None, :, None, None]
    ref = ref.permute(0, 2, 3, 1)
    outs = {}
    for la in ("0",):
        for seed in (1, 2, 3):
            emu.config(cq, dma_late=1, greedy=1, seed=seed)
            c0 = emu.counters(cq)
            outs[(la, seed)] = emu.conv_fwd(cq, x, w, 3, 3, 1, relu_in=relu, up=up, pool=pool, bias=bias, alpha=0.25 if pool else 1.0,      # (the pooling epilogue SUMS the quad)
                                            env=env).copy()
            assert emu.counters(cq)["dma_ops"] > c0["dma_ops"]
    got = _t64(outs[("0", 1)])
    assert (got - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-6, case
    for k, v in outs.items():
        assert np.array_equal(v, outs[("0", 1)]), (case, k)


# ---- the other forward engines behind sg_conv2d_fwd, pinned for later edits ---------------------------------------------------------------------------------
ENGINE_CASES = [
    # name, env, (N, H, C, Cout, R, pad, relu_in, up, pool)
    ("conv_v3 deep layer", {"SG_CONV_V3": "force", "SG_CONV_V4": "0", "SG_QUAD": "0"}, (2, 8, 512, 192, 3, 1, True, False, False)),
    ("conv_v3 pooled", {"SG_CONV_V3": "force", "SG_CONV_V4": "0", "SG_QUAD": "0"}, (2, 8, 416, 96, 3, 1, True, False, True)),
    ("conv_sk 1x1", {"SG_CONV_SK": "force"}, (2, 16, 96, 48, 1, 0, False, False, False)),
    ("conv_sk 1x1 upsampled, ReLU", {"SG_CONV_SK": "force"}, (2, 8, 192, 96, 1, 0, True, True, False)),
    ("conv_sk stem 3x3 on 8 channels", {"SG_CONV_SK": "force", "SG_CONV_RS": "0"}, (2, 32, 8, 96, 3, 1, False, False, False)),
    ("conv_rs RGB layer (8 padded couts, 128-wide rows)", {"SG_CONV_RS": "force", "SG_CONV_RS96": "0"}, (1, 128, 96, 8, 3, 1, True, False, False)),
    ("conv_rs96 row streaming 96 -> 96", {"SG_CONV_RS": "force", "SG_CONV_RS96": "force"}, (1, 128, 96, 96, 3, 1, False, False, False)),
    ("conv_v2 tile kernel, 5x5", {"SG_CONV_V2": "force", "SG_CONV_V3": "0", "SG_CONV_V4": "0"}, (2, 16, 24, 96, 5, 2, False, False, False)),
    ("generic engine (everything off)", {"SG_CONV_V2": "0", "SG_CONV_V3": "0", "SG_CONV_V4": "0", "SG_CONV_SK": "0", "SG_CONV_RS": "0", "SG_QUAD": "0"},
     (2, 8, 40, 24, 3, 1, True, False, False)),
]


@pytest.mark.parametrize("case", ENGINE_CASES, ids=[c[0] for c in ENGINE_CASES])
def test_forward_engines_pin_interpreter(cq, case):
    """every forward / data-gradient engine of sg_conv2d_fwd against torch on the same bf16 inputs (one bf16 rounding of the result), under late DMA
    completion and a seeded wave order: the regression net for edits made without GPU time"""
    import torch
    import torch.nn.functional as TF
    name, env, (N, H, Cin, Cout, R, pad, relu, up, pool) = case
    rng = np.random.default_rng(51)
    x = emu.to_bf16(rng.standard_normal((N, H, H, Cin)).astype(np.float32))
    w = emu.to_bf16((0.1 * rng.standard_normal((Cout, R, R, Cin))).astype(np.float32))
    bias = rng.standard_normal(Cout).astype(np.float32)
    xr = _t64(x).permute(0, 3, 1, 2)
    if relu:
        xr = xr.clamp(min=0)
    if up:
        xr = TF.interpolate(xr, scale_factor=2, mode="nearest")
    ref = TF.conv2d(xr, _t64(w).permute(0, 3, 1, 2), None, padding=pad)
    if pool:
        ref = TF.avg_pool2d(ref, 2)
    ref = (ref + torch.from_numpy(bias).double()[None, :, None, None]).permute(0, 2, 3, 1)
    for seed in (1, 2):
        emu.config(cq, dma_late=1, greedy=1, seed=seed)
        out = emu.conv_fwd(cq, x, w, R, R, pad, relu_in=relu, up=up, pool=pool, bias=bias, alpha=0.25 if pool else 1.0, env=env)
        assert (_t64(out) - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-6, (name, seed)


WGRAD_ENGINE_CASES = [
    # name, env, (N, H, C, Cout, R, pad, relu)
    ("wgrad_sk 1x1", {"SG_WGRAD_SK": "1", "SG_WGRAD_V3": "0"}, (2, 16, 96, 48, 1, 0, True)),
    ("wgrad_sk stem 3x3 on 8 channels", {"SG_WGRAD_SK": "1", "SG_WGRAD_V3": "0"}, (2, 32, 8, 96, 3, 1, False)),
    ("wgrad_v2 tile kernel", {"SG_CONV_V2": "force", "SG_WGRAD_V3": "0", "SG_WGRAD_SK": "0"}, (2, 16, 64, 128, 3, 1, True)),
    ("generic engine", {"SG_CONV_V2": "0", "SG_WGRAD_V3": "0", "SG_WGRAD_SK": "0"}, (2, 8, 40, 24, 3, 1, False)),
]


@pytest.mark.parametrize("case", WGRAD_ENGINE_CASES, ids=[c[0] for c in WGRAD_ENGINE_CASES])
def test_wgrad_engines_pin_interpreter(wg, case):
    """the other weight-gradient engines of sg_conv2d_wgrad (streaming thin-layer kernel, LDS-DMA tile kernel, generic contraction) against the fp64 formula"""
    name, env, (N, H, Cin, Cout, R, pad, relu) = case
    rng = np.random.default_rng(61)
    x = emu.to_bf16(rng.standard_normal((N, H, H, Cin)).astype(np.float32))
    dy = emu.to_bf16(rng.standard_normal((N, H, H, Cout)).astype(np.float32))
    xf = emu.PIX_RELU if relu else 0
    emu.config(wg, dma_late=1, greedy=1, seed=4)
    dw, _, _ = emu.conv_wgrad(wg, x, dy, Cout, x_flags=xf, env=env, R=R, pad=pad)
    ref, _ = emu.wgrad_ref(x, dy, x_flags=xf, R=R, pad=pad)
    assert np.abs(dw - ref).max() <= 3e-6 * np.abs(ref).max(), name

