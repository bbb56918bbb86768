"""Whole training steps of the package on the CPU: its Python layer (autograd functions, weight bank, fused optimiser, worker) drives its real launch
sequence through libsgamd's own kernel SOURCES, interpreted lane by lane (tests/hipemu/fullemu.py), and the result is held against the golden
fixtures the real reference wrote (tests/golden/*.npz) -- the same check tests/test_model_gpu.py::test_training_step_vs_golden makes on the GPU, same
tolerances. What this is for: host-side changes (launch order, fused operands, weight-bank layouts) can be checked at network level without
GPU time. What it is not: a product path (the package is bound to the interpreter by this test process only) or a performance statement.

One fixture runs by default (an fp32 BigGAN step at width 8: ~10 s after the one-off host build of the library, ~1 min on 8 cores); SG_EMU_NET=1 runs every width-8
fixture in fp32 (tools/sessions/cpu_emu_nets.sh; last full run: profiles/r04_hipemu_nets.txt)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
sys.path.insert(0, os.path.dirname(HERE))
import emu  # noqa: E402

pytestmark = pytest.mark.skipif(not emu.available(), reason="host clang++ of the ROCm toolchain not found")
FULL = os.environ.get("SG_EMU_NET") == "1"
ALL = ["dcgan32", "sndcgan32", "sngan32", "resgan32", "wgangp32", "sngp32", "bigdeep32", "bigdeepsg32"]


@pytest.fixture(autouse=True)
def _one_torch_thread():
    # torch's OpenMP workers keep spinning after a parallel region: next to the interpreter's own thread pool they cost a factor of ~7 in wall time
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _step(name, mixed):
    import fullemu
    import test_model_gpu as TM
    import contextlib
    import io
    log = io.StringIO()
    with fullemu.Installed(dma_late=1, greedy=1, seed=1) as E:
        c0 = E.counters()
        try:
            with contextlib.redirect_stdout(log):          # (the per-tensor report of the comparison: shown only when it fails)
                TM.step_vs_golden(name, mixed, torch.device("cpu"))
        except BaseException:
            print(log.getvalue())
            raise
        c1 = E.counters()
    assert c1["launches"] - c0["launches"] > 100 and c1["mfma"] > c0["mfma"]        # the kernels ran (there is nothing else that could have)
    return {k: c1[k] - c0[k] for k in c1}


def test_emulated_training_step_vs_golden_default():
    """one fixture in the default CPU suite: an fp32 BigGAN step at width 8 (~700 launches: generic GEMM convolutions forward / data / weight gradient,
    spectral norm forward + backward, conditional batch norm, self-attention, projection discriminator, hinge loss, fused Adam + EMA), the reference's
    golden vectors at the fp32 tolerances"""
    _step("biggan32", False)


@pytest.mark.skipif(not FULL, reason="SG_EMU_NET=1 runs every width-8 fixture through the interpreter (minutes)")
@pytest.mark.parametrize("name", ALL)
def test_emulated_fp32_training_step_vs_golden(name):
    _step(name, False)


def test_emulated_grouped_cbn_affine_equals_per_layer():
    """functional.cbn_prefetch / CbnAffineGroupFn (csrc/linear_group.hip: every conditional batch norm's [1 + gain(y) | bias(y)] rows of a generator forward in one
    launch) against the per-layer GEMMs it replaces: the generated images, the gradient w.r.t. z and every parameter gradient of the generator (the conditional
    batch norms' linears among them) of the biggan32 fixture, fp32; and the launch really happened"""
    import fullemu
    import test_model_gpu as TM
    from util import load_golden, sub
    from studiogan_amd import functional as F
    dev = torch.device("cpu")
    fix, meta = load_golden("biggan32")
    y = meta["yaml"]
    ins = {k: v.to(dev) for k, v in sub(fix, "in/").items()}
    res, calls = {}, {True: 0, False: 0}
    with fullemu.Installed(dma_late=1, greedy=1, seed=3) as E:
        call0 = E.L.call
        for grouped in (True, False):
            F._CBN_GROUP[0] = grouped

            def counting(name, *a, _g=grouped):
                calls[_g] += name == "sg_linear_group"
                return call0(name, *a)
            E.L.call = counting
            try:
                G, _ = TM.build_from_yaml(y, False, dev)
                G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
                G.train()
                z = ins["z0"].clone().requires_grad_(True)
                img = G(z, ins["fl0"])
                (img * torch.linspace(-1, 1, img.numel()).view_as(img)).sum().backward()
                grads = {k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None}
                res[grouped] = (img.detach().clone(), z.grad.clone(), grads)
            finally:
                F._CBN_GROUP[0] = True
                E.L.call = call0
    assert calls == {True: 1, False: 0}
    a, b = res[True], res[False]
    assert torch.allclose(a[0], b[0], rtol=0, atol=2e-5 * b[0].abs().max().item())      # (one FMA chain against the f32 MFMA's summation: fp32 rounding, 2e-6 measured)
    assert torch.allclose(a[1], b[1], rtol=0, atol=2e-4 * b[1].abs().max().item() + 1e-12)
    assert a[2].keys() == b[2].keys() and any(".bn1.gain." in k for k in a[2])
    for k in a[2]:
        if ".conv2d" in k and k.endswith(".bias"):      # a bias in front of a batch norm: its true gradient is zero, what is there is rounding noise
            continue
        assert torch.allclose(a[2][k], b[2][k], rtol=0, atol=2e-4 * b[2][k].abs().max().item() + 1e-12), k


def test_emulated_batch_norm_counters_in_one_launch():
    """ops.bump_batches_tracked: a generator forward in training mode moves every BatchNorm2d.num_batches_tracked by one (one add over the int64 words of the
    buffer arena); with one batch norm frozen the modules fall back to their own increments and the frozen one stays; the EMA twin's buffer update copies them"""
    import copy
    import fullemu
    import test_model_gpu as TM
    from util import load_golden, sub
    from studiogan_amd import ops, optim
    dev = torch.device("cpu")
    fix, meta = load_golden("biggan32")
    ins = {k: v.to(dev) for k, v in sub(fix, "in/").items()}
    with fullemu.Installed(dma_late=1, greedy=1, seed=4):
        G, _ = TM.build_from_yaml(meta["yaml"], False, dev)
        G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
        G.train()
        bns = [m for m in G.modules() if isinstance(m, ops.BatchNorm2d)]
        assert len(bns) >= 7
        base = [int(m.num_batches_tracked) for m in bns]
        with torch.no_grad():
            G(ins["z0"], ins["fl0"])
            G(ins["z0"], ins["fl0"])
        assert [int(m.num_batches_tracked) for m in bns] == [b + 2 for b in base]
        assert not ops._NBT_BULK[0]
        bns[2].eval()
        with torch.no_grad():
            G(ins["z0"], ins["fl0"])
        assert [int(m.num_batches_tracked) for m in bns] == [b + (2 if i == 2 else 3) for i, b in enumerate(base)]
        bns[2].train()
        Ge = copy.deepcopy(G)
        with torch.no_grad():
            G(ins["z0"], ins["fl0"])
        ema = optim.Ema(G, Ge, 0.9, 0)
        ema.update_buffers(0.9)
        assert [int(m.num_batches_tracked) for m in Ge.modules() if isinstance(m, ops.BatchNorm2d)] == [int(m.num_batches_tracked) for m in bns]
        assert sorted(G.state_dict().keys()) == sorted(Ge.state_dict().keys())


def test_emulated_frozen_network_weight_image_cache():
    """bank.WeightBank.begin_forward keeps the emitted weight images of a frozen network (eval mode, no graph: the evaluation generator of the FID
    loop) until something writes its parameters: hits give bit-identical images, a torch-side write (in-place op, load_state_dict), a raw-pointer
    write (the fused Adam + EMA launch) and a training-mode forward (power iteration) each force a re-emission that equals the uncached one."""
    import copy
    import fullemu
    import test_model_gpu as TM
    from util import load_golden, sub, hyper
    from studiogan_amd import bank as B, ops
    from studiogan_amd.worker import Worker
    dev = torch.device("cpu")
    fix, meta = load_golden("biggan32")
    y = meta["yaml"]
    with fullemu.Installed(dma_late=1, greedy=1, seed=2) as E:
        G, D = TM.build_from_yaml(y, False, dev)
        G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
        D.load_state_dict({k: v.to(dev) for k, v in sub(fix, "D_init/").items()}, strict=True)
        opt = hyper(y)
        w = Worker(G, D, opt["z_dim"], y["DATA"]["num_classes"], meta["batch"], opt["adv_loss"], opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"],
                   d_updates_per_step=1, apply_g_ema=True, g_ema_decay=0.9, g_ema_start=0, apply_gp=opt["apply_gp"], gp_lambda=opt["gp_lambda"])
        ins = {k: v.to(dev) for k, v in sub(fix, "in/").items()}
        z, lab = ins["z0"], ins["fl0"]
        Ge = w.Gen_ema
        Ge.eval()
        _, bank = ops._root_and_bank(Ge)

        def run(cache=True):
            B._EVAL_CACHE[0] = cache                  # (opt-in in the package: SG_EVAL_CACHE=1)
            try:
                with torch.no_grad():
                    Ge(z, lab, eval=True)
                s0 = bank.slots[0]      # what the cache keeps: the emitted operand images of the no-graph slot (the width-8 generator's tanh output is saturated)
                return torch.cat([s0.img.float().flatten(), s0.f32.flatten()]).clone()
            finally:
                B._EVAL_CACHE[0] = False

        def same(u, v):      # two emissions of the same weights: equal up to the order of the fp32 atomics in the spectral-norm reductions
            ok = torch.allclose(u, v, rtol=1e-5, atol=2e-6, equal_nan=True)        # (padding the emission never writes may hold anything)
            if not ok:
                d = (u - v).abs()
                print("emissions differ: max", d.max().item(), "at", d.argmax().item(), u[d.argmax()].item(), v[d.argmax()].item(), "n", (d > 2e-6).sum().item())
            return ok

        def hits():
            return bank.__dict__.get("eval_cache_hits", 0)
        a = run()
        h0, l0 = hits(), E.counters()["launches"]
        b = run()
        l1 = E.counters()["launches"]
        c = run()
        l2 = E.counters()["launches"]
        assert hits() == h0 + 2 and torch.equal(a, b) and torch.equal(a, c)
        assert l2 - l1 == l1 - l0                                         # (steady state: the same launches per cached forward)
        ref = run(cache=False)
        l3 = E.counters()["launches"]
        assert same(ref, a) and (l3 - l2) > (l2 - l1)              # the uncached forward issues the spectral-norm / packing launches on top
        # 1. torch-side write
        p = next(q for n, q in Ge.named_parameters() if n.endswith("weight_orig") or n.endswith(".weight"))
        with torch.no_grad():
            p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(5)))
        h = hits()
        d = run()
        assert hits() == h and not torch.equal(d, a) and same(d, run(cache=False))
        # 2. raw-pointer write: one generator update moves G and, inside the same launch, the EMA copy
        w.train_generator(0, [(ins["z1"], ins["fl1"])])
        Ge.eval()                                     # (the worker leaves its networks in training mode)
        h = hits()
        e = run()
        assert hits() == h and not torch.equal(e, d) and same(e, run(cache=False))
        e2 = run()                                    # (the uncached emission above left the slot unkeyed: this one re-emits and keys it)
        assert same(e2, e) and torch.equal(run(), e2) and hits() == h + 1
        # 3. load_state_dict
        Ge.load_state_dict(copy.deepcopy(G.state_dict()), strict=True)
        h = hits()
        f = run()
        assert hits() == h and not torch.equal(f, e) and same(f, run(cache=False))
        # 4. a training-mode forward of the same network runs the power iteration (u / v move): the next frozen forward re-emits
        Ge.train()
        with torch.no_grad():
            Ge(z, lab)
        Ge.eval()
        h = hits()
        g2 = run()
        assert hits() == h and same(g2, run(cache=False))


# ---- the FAST kernels at full width (SG_EMU_NET=1: 1-3 min each) ------------------------------------------------------------------------------------------
FORCE = ("SG_CONV_V4", "SG_CONV_V3", "SG_CONV_V2", "SG_CONV_SK", "SG_CONV_RS", "SG_CONV_RS96", "SG_WGRAD_BJ256", "SG_WGRAD_V3")


@pytest.fixture
def forced_fast_kernels(monkeypatch):
    """what tests/test_fullwidth_gpu.py::forced does: the kernels bench.py runs at batch 256 take the fixtures' small batches too"""
    for k in FORCE:
        monkeypatch.setenv(k, "force")


@pytest.mark.skipif(not FULL, reason="SG_EMU_NET=1: BigGAN-128 at FULL width (batch 4) through the interpreter, ~3 min")
@pytest.mark.parametrize("lean", ["0", "1"])
def test_emulated_fullwidth_bf16_step_vs_golden(forced_fast_kernels, monkeypatch, lean):
    """the benchmarked network (BASELINE config 3: BigGAN, ImageNet-128 widths) with the benchmarked kernels -- conv_q / conv_v4 / conv_v3 / conv_sk /
    conv_rs forward and data gradient, wgrad_v3 / wgrad_q / wgrad_sk / wgrad_v2, flash attention, epilogue BN statistics: 901 launches, 52 M MFMAs --
    one bf16 training step against the reference's golden vectors. lean = 1: the default weight-gradient kernels (wgrad_v3l.h / wgrad_ql.h), 0: the round-4 ones."""
    monkeypatch.setenv("SG_WGRAD_V3_LEAN", lean)
    monkeypatch.setenv("SG_WGRAD_Q_LEAN", lean)
    c = _step("biggan128w", True)
    assert c["dma_ops"] > 1e6 and c["tr_reads"] > 1e6          # the LDS-DMA / transpose-read kernels ran (the width-8 fixtures never reach them)


@pytest.mark.skipif(not FULL, reason="SG_EMU_NET=1: teacher-forced block test of BigGAN-128 at full width through the interpreter, 1-2 min each")
@pytest.mark.parametrize("which", ["D", "G"])
def test_emulated_fullwidth_teacher_forced_blocks_with_lean_kernels(forced_fast_kernels, monkeypatch, which):
    """tests/test_blocks_gpu.py::bf16_vs_emulating_oracle on the interpreter with the lean weight-gradient kernels (the default since round 5): every block of the full-width
    network on the bf16-emulating oracle's input and upstream gradient -- block output, input gradient and every weight gradient to 1e-2 relative L2
    (discriminator: flat; generator: max(1e-2, 1.5 x measured floor)). The tight network-level bound on the lean weight-gradient kernels."""
    import contextlib
    import io
    import fullemu
    import test_blocks_gpu as TB
    monkeypatch.setenv("SG_WGRAD_V3_LEAN", "1")
    monkeypatch.setenv("SG_WGRAD_Q_LEAN", "1")
    rows, log = [], io.StringIO()
    with fullemu.Installed(dma_late=1, greedy=1, seed=1):
        try:
            with contextlib.redirect_stdout(log):
                TB.bf16_vs_emulating_oracle("biggan128w", which, report=rows, dev=torch.device("cpu"))
        except BaseException:
            print(log.getvalue()[-6000:])
            raise
    worst = max((e for n, e, t in rows if "teacher-forced grad" in n), default=0.0)
    print(f"{which}: worst teacher-forced weight-gradient error {worst:.3e} over {len(rows)} compared tensors")
    assert rows and worst < 1e-2 * (1 if which == "D" else 3)


@pytest.mark.skipif(not FULL, reason="SG_EMU_NET=1: every full-width fixture through the interpreter, 0.5-4 min each")
@pytest.mark.parametrize("name", ["sngan32w", "wgangp128w", "bigdeep128w", "biggan128w"])
def test_emulated_fullwidth_bf16_step_every_fixture(forced_fast_kernels, monkeypatch, name):
    """the four full-width fixtures (C2 SNGAN, C5 WGAN-GP ResNet-128 with its double backward, C4 BigGAN-deep-128, C3 BigGAN-128) with the fast kernels forced:
    golden vectors of the reference at the bf16 tolerances of tests/test_fullwidth_gpu.py."""
    _step(name, True)


@pytest.mark.parametrize("name,kind", [("biggan32", "r1"), ("biggan32", "maxgp"), ("bigdeep32", "r1")])
def test_emulated_second_order_through_attention(name, kind):
    """R1 / max-gradient penalty on the BigGAN discriminator (SelfAttention with sigma = 0.6; reference utils/losses.py:338-361 through utils/ops.py:83-103): the
    create_graph pass through the attention block (functional.AttnPooledFn / AttnOutFn / MaxPool2Fn with BmmFn, SoftmaxRowsFn, ScalePtrFn and the csrc/ext
    adjoints) against torch autograd's double backward over the CPU oracle -- penalty and every parameter gradient (tests/test_blocks_gpu.py r1_maxgp_case)"""
    import fullemu
    import test_blocks_gpu as TB
    with fullemu.Installed(dma_late=1, greedy=1, seed=4) as E:
        c0 = E.counters()
        TB.r1_maxgp_case(name, kind, torch.device("cpu"))      # (bigdeep32: + the bottleneck blocks' channel-concat skip, functional.CatConvDgradFn)
        assert E.counters()["launches"] - c0["launches"] > 100


def test_emulated_run_config_tool(tmp_path, monkeypatch, capsys):
    """tools/run_config.py: a configuration file in the reference's YAML layout -> config_map.build -> two training steps (two discriminator updates each, R1 +
    DiffAugment + EMA on a spectral-norm ResNet with a projection head) on the interpreter; one JSON line with finite losses."""
    import json
    import yaml
    cfg = {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
           "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "g_conv_dim": 8, "d_conv_dim": 8, "z_dim": 32,
                     "apply_g_ema": True, "g_ema_decay": 0.9999, "g_ema_start": 0},
           "LOSS": {"adv_loss": "hinge", "apply_r1_reg": True, "r1_lambda": 1.0, "r1_place": "inside_loop"},
           "AUG": {"apply_diffaug": True, "diffaug_type": "diffaug"},
           "OPTIMIZATION": {"batch_size": 4, "d_updates_per_step": 2, "beta1": 0.0, "beta2": 0.9}}
    f = tmp_path / "Tiny-R1-DiffAug.yaml"
    f.write_text(yaml.safe_dump(cfg))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import run_config
    monkeypatch.setattr(sys, "argv", ["run_config.py", str(f), "--emulate", "--steps", "1", "--warmup", "1"])
    run_config.main()
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["backbone"] == "resnet" and out["batch"] == 4 and out["device"] == "interpreter"
    assert out["d_loss"] == out["d_loss"] and out["g_loss"] == out["g_loss"] and abs(out["d_loss"]) < 1e4


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/configs"), reason="the reference checkout is only present in the authoring container")
def test_emulated_architecture_parity_against_the_reference_in_fp64(monkeypatch, capsys):
    """tools/config_parity_emulated.py on two configuration files (the sweep over all 145: profiles/r05_config_parity_emulated.txt): the REAL reference's networks, run in
    fp64, against this package's built through config_map from the same file and loaded with the same state -- image, discriminator outputs, every first-order
    parameter gradient and the R1 penalty with ITS parameter gradients (double backward), kernels on the interpreter."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import config_parity_emulated as T
    monkeypatch.setattr(sys, "argv", ["config_parity_emulated.py", "--dir=CIFAR10", "--batch=4", "--r1", "SNGAN", "ReACGAN-TAC"])
    T.main()
    out = capsys.readouterr().out
    rows = [ln for ln in out.splitlines() if ln.startswith(("SNGAN ", "ReACGAN-TAC "))]
    assert len(rows) == 2 and all(" ok " in r and "MISMATCH" not in r for r in rows), out
    assert "2 distinct architectures agree" in out


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/configs"), reason="the reference checkout is only present in the authoring container")
def test_emulated_training_step_against_the_references_own_worker(monkeypatch, capsys):
    """tools/config_worker_parity_emulated.py on two configuration files (the sweep over all of them: profiles/r05_config_worker_parity_emulated.txt): the reference's
    UNMODIFIED WORKER.train_discriminator / train_generator (src/worker.py:213-681), run on the CPU from the reference's own constructor, against this package's Worker
    from the same state, real batches and torch seed -- both consume the generator identically, no draw is injected: two discriminator updates (Adam in between) and one
    generator update of SNGAN + DiffAugment + LeCam and of ReACGAN (D2D-CE head) + adaptive augmentation, kernels on the interpreter."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import config_worker_parity_emulated as T
    monkeypatch.setattr(sys, "argv", ["config_worker_parity_emulated.py", "--dir=CIFAR10", "--batch=4", "SNGAN-DiffAug-LeCam", "ReACGAN-ADA"])
    T.main()
    out = capsys.readouterr().out
    rows = [ln for ln in out.splitlines() if ln.startswith(("SNGAN-DiffAug-LeCam ", "ReACGAN-ADA "))]
    assert len(rows) == 2 and all(r.rstrip().split("  ")[-2].strip().endswith("ok") or " ok " in r for r in rows) and not any("MISMATCH" in r or "FAILED" in r for r in rows), out
    assert "2 configuration files" in out


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/configs"), reason="the reference checkout is only present in the authoring container")
def test_config_steps_fixture_regenerates_from_the_reference(tmp_path, monkeypatch, capsys):
    """tests/golden/config_steps.npz: two of its 55 files re-emitted by tools/config_worker_parity_emulated.py from the reference's own worker -- the recorded draws are
    bit-identical to the committed ones, the losses and gradient norms agree to the last digits a re-run of torch's CPU kernels keeps"""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import config_worker_parity_emulated as T
    out = str(tmp_path / "steps")
    monkeypatch.setattr(sys, "argv", ["config_worker_parity_emulated.py", "--dir=CIFAR10", "--batch=4", "--emit=" + out, "SNGAN-DiffAug", "ContraGAN-TAC"])
    T.main()
    capsys.readouterr()
    a, b = np.load(os.path.join(HERE, "golden", "config_steps.npz")), np.load(out + ".npz")
    assert len(b.files) > 10
    for k in b.files:
        if "/draw" in k:
            assert np.array_equal(a[k], b[k]), k
        else:
            assert np.allclose(a[k], b[k], rtol=1e-5, atol=0), (k, a[k], b[k])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_emulated_sn_kernels_against_torch_spectral_norm(dtype):
    """tests/test_sn_gpu.py's kernel-level spectral-norm table (sg_sn_forward / sg_sn_backward against torch.nn.utils.spectral_norm in fp64) on the interpreter"""
    import fullemu
    import sn_checks as SC
    with fullemu.Installed(dma_late=1, greedy=1, seed=2) as E:
        L = E.L
        rows = SC.run(torch.device("cpu"), dtype, L, L.call, L.ptr, L.stream)
    bad = [(n, e, t) for n, e, t in rows if not e <= t]
    assert not bad, bad


def test_emulated_sn_forward_table_longer_than_one_flat_tile_table():
    """tests/test_sn_gpu.py's 70-layer table (csrc/sn.hip walks it in runs of SN_MAXL = 64) on the interpreter: one call = two calls on the halves, bit for bit"""
    import fullemu
    import sn_checks as SC
    shapes = [(16 + 8 * (i % 3), 8 + 8 * (i % 2), 3 if i % 5 else 1) for i in range(70)]
    with fullemu.Installed(dma_late=1, greedy=1, seed=3) as E:
        L = E.L
        one = SC.forward_table(torch.device("cpu"), torch.bfloat16, L, L.call, L.ptr, L.stream, shapes, seed=5)
        two = SC.forward_table(torch.device("cpu"), torch.bfloat16, L, L.call, L.ptr, L.stream, shapes, seed=5, split_at=37)
    for i, (x, y) in enumerate(zip(one, two)):
        for name, p, q in zip(("u", "v", "sigma", "w_fwd", "w_dgrad"), x, y):
            assert torch.equal(p, q), (i, shapes[i], name)
    assert all(float(x[2]) > 0 for x in one)


@pytest.mark.parametrize("case", [(2, 64, 96, 9, 9, 3, 3, 1, (1, 1)), (1, 128, 160, 9, 9, 1, 7, 1, (0, 3)), (2, 48, 32, 7, 7, 5, 5, 2, (2, 2))])
def test_emulated_conv_fwd_f32_bf16x3_split(case):
    """the generic engine's "bf16x3" fp32 mode (csrc/gemm_core.h SPLIT: fp32 operands split into two bf16 terms at fragment time, three bf16 MFMAs per k-tile) on
    the interpreter against F.conv2d in fp64, next to the exact fp32 MFMA path (tests/test_kernels_gpu.py::test_conv_fwd_f32_bf16x3_split on the GPU)"""
    import fullemu
    import test_kernels_gpu as TK
    with fullemu.Installed(dma_late=1, greedy=1, seed=4):
        e_exact, e_split = TK.f32_split_case(torch.device("cpu"), case, sync=lambda: None)
        w_exact, w_split = TK.f32_split_wgrad_case(torch.device("cpu"), case, sync=lambda: None)
    assert e_exact <= 2e-6 and e_split <= 2e-5 and e_split > e_exact, (e_exact, e_split)
    assert w_exact <= 4e-6 and w_split <= 2e-5 and w_split > w_exact, (w_exact, w_split)
