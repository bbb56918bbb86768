"""CPU: host-side logic that needs no GPU -- flat arenas, state_dict naming, the C ABI surface."""
import copy
import os
import re
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_arena_views_and_grad_views():
    from studiogan_amd.bank import ParamArena, arena_of, ensure_grad, get_buffer_arena
    net = nn.Sequential(nn.Linear(5, 7), nn.BatchNorm1d(7), nn.Linear(7, 3))
    ref = [p.detach().clone() for p in net.parameters()]
    a = ParamArena(list(net.parameters()))
    assert a.intact()
    for p, r in zip(net.parameters(), ref):
        assert torch.equal(p.detach(), r)
        ent = arena_of(p)
        assert ent is not None and ent[0] is a
        assert p.data_ptr() == a.data.data_ptr() + 4 * ent[1]
        g = ensure_grad(p)
        assert g.data_ptr() == a.grad.data_ptr() + 4 * ent[1] and float(g.abs().sum()) == 0.0
    # arena values follow in-place parameter updates and vice versa
    with torch.no_grad():
        net[0].weight.add_(1.0)
    assert torch.equal(a.data[:35].view(7, 5), net[0].weight.detach())
    # buffers
    rm = net[1].running_mean
    ba = get_buffer_arena(net)
    assert net[1].running_mean is rm and ba.intact(net)
    net(torch.randn(4, 5))
    assert torch.equal(ba.data[ba.offsets[0]:ba.offsets[0] + 7], net[1].running_mean)
    # deepcopy detaches from the arenas and builds its own lazily
    net2 = copy.deepcopy(net)
    assert get_buffer_arena(net2) is not ba
    assert arena_of(next(net2.parameters())) is None


def test_abi_symbols_match_header():
    """libsgamd.so loads and exports every function include/sgamd.h declares (no compute calls without a GPU)."""
    import studiogan_amd
    from studiogan_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sgamd.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = studiogan_amd.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in sgamd.h but not exported"
    assert declared == set(_lib.exported_symbols()), (declared ^ set(_lib.exported_symbols()))
    assert lib.sg_version() >= 1


def test_struct_layouts_match_c():
    """ctypes mirrors of the descriptor structs have the sizes the C compiler gives them."""
    import ctypes
    import subprocess
    import tempfile
    from studiogan_amd import _lib
    src = ('#include <stdio.h>\n#include "sgamd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(sg_conv_fwd_desc), '
           'sizeof(sg_conv_wgrad_desc), sizeof(sg_gemm_desc), sizeof(sg_sn_layer), sizeof(sg_sn_bwd_layer), sizeof(sg_conv_skip_desc));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout.split()
    sizes = [ctypes.sizeof(x) for x in (_lib.ConvFwdDesc, _lib.ConvWgradDesc, _lib.GemmDesc, _lib.SnLayer, _lib.SnBwdLayer, _lib.ConvSkipDesc)]
    assert [int(v) for v in out] == sizes


def test_module_names_and_sn_state():
    from studiogan_amd import ops
    m = ops.snconv2d(8, 16, 3, 1, 1)
    sd = m.state_dict()
    assert set(sd.keys()) == {"weight_orig", "weight_u", "weight_v", "bias"}
    assert sd["weight_orig"].shape == (16, 8, 3, 3) and sd["weight_u"].shape == (16,) and sd["weight_v"].shape == (72,)
    assert abs(float(sd["weight_u"].norm()) - 1) < 1e-5
    assert isinstance(m, nn.Conv2d)
    e = ops.sn_embedding(10, 6)
    assert set(e.state_dict().keys()) == {"weight_orig", "weight_u", "weight_v"} and isinstance(e, nn.Embedding)
    lin = ops.linear(4, 5)
    assert set(lin.state_dict().keys()) == {"weight", "bias"} and isinstance(lin, nn.Linear)
    bn = ops.batchnorm_2d(6)
    assert bn.eps == 1e-4 and bn.momentum == 0.1 and isinstance(bn, nn.modules.batchnorm._BatchNorm)
    # orthogonal init reaches weight_orig (reference init_weights goes through module.weight's shared storage)
    ops.init_weights(lambda: [m], "ortho")
    w = m.weight_orig.detach().reshape(16, -1)
    assert torch.allclose(w @ w.t(), torch.eye(16), atol=1e-4)


@pytest.mark.parametrize("name", ["biggan32", "sngan32", "resgan32", "dcgan32", "sndcgan32", "bigdeep32", "bigdeepsg32", "sngan32w", "wgangp128w"])
def test_golden_state_dict_keys_match_backbone(name):
    from util import load_golden, sub
    from test_model_gpu import build_from_yaml
    fix, meta = load_golden(name)
    G, D = build_from_yaml(meta["yaml"], False, torch.device("cpu"))
    assert set(G.state_dict().keys()) == set(sub(fix, "G_init/").keys())
    assert set(D.state_dict().keys()) == set(sub(fix, "D_init/").keys())
    G.load_state_dict(sub(fix, "G_init/"), strict=True)
    D.load_state_dict(sub(fix, "D_init/"), strict=True)
    with pytest.raises(RuntimeError):
        G(torch.randn(2, G.z_dim), torch.zeros(2, dtype=torch.long))  # CPU tensors: no fallback on the product path


def test_synthetic_inception_weights_match_oracle_generator():
    """bench.py's FID leg takes its seeded random Inception weights from the product package (the oracle is test infrastructure and is
    not imported by anything that is measured); both generators must produce the same tensors for the same seed."""
    from studiogan_amd import metrics as M
    from oracle import inception as OI
    a, b = M.synthetic_state_dict(3), OI.random_state_dict(3)
    assert list(a.keys()) == list(b.keys())
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_loss_schedule_helpers():
    """reference src/utils/losses.py:364-366 (adjust_k) and src/utils/ops.py:106-133 (LeCamEMA)."""
    from studiogan_amd import losses as SL, ops
    k = 64
    for _ in range(500):
        k = SL.adjust_k(current_k=k, topk_gamma=0.99, inf_k=int(64 * 0.5))
    assert k == 32
    e = ops.LeCamEMA(decay=0.9, start_iter=10)
    assert e.D_real == 7777
    e.update(2.0, "D_real", 3)            # before start_iter: decay 0
    assert e.D_real == 2.0
    e.update(4.0, "D_real", 10)
    assert abs(e.D_real - (2.0 * 0.9 + 4.0 * 0.1)) < 1e-12
    with pytest.raises(ValueError):
        e.update(1.0, "nope", 0)


def test_fused_adam_checkpoint_layout_roundtrip():
    """FusedAdam.state_dict()/load_state_dict() interchange with torch.optim.Adam's checkpoint layout (the reference saves
    optimizer.state_dict() and restores it with load_state_dict, src/utils/ckpt.py): exp_avg / exp_avg_sq / step survive both ways."""
    from studiogan_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(5, 7), nn.Linear(7, 3))
    ref = copy.deepcopy(net)
    ropt = torch.optim.Adam(ref.parameters(), lr=2e-4, betas=(0.0, 0.999), eps=1e-6)
    for _ in range(3):
        ropt.zero_grad()
        ref(torch.randn(4, 5)).square().sum().backward()
        ropt.step()
    sd = ropt.state_dict()
    opt = FusedAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.9), eps=1e-6)
    opt.load_state_dict(copy.deepcopy(sd))
    assert opt._t == 3
    g = opt.param_groups[0]
    assert g["lr"] == 2e-4 and tuple(g["betas"]) == (0.0, 0.999)
    a = opt._arena
    for p, o, q in zip(a.params, a.offsets, ref.parameters()):
        st = ropt.state[q]
        assert torch.equal(opt._m[o:o + p.numel()].view(p.shape), st["exp_avg"])
        assert torch.equal(opt._v[o:o + p.numel()].view(p.shape), st["exp_avg_sq"])
    out = opt.state_dict()
    assert out["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    assert set(out["state"].keys()) == set(sd["state"].keys())
    for i in sd["state"]:
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(out["state"][i][k], sd["state"][i][k])
        assert float(out["state"][i]["step"]) == float(sd["state"][i]["step"]) == 3.0
    # and torch.optim.Adam accepts what FusedAdam wrote (a StudioGAN checkpoint written by this package resumes on the reference)
    ropt2 = torch.optim.Adam(copy.deepcopy(ref).parameters(), lr=1.0)
    ropt2.load_state_dict(out)
    assert len(opt.state) == 0, "the flat arenas stay the single source of truth"


def test_weight_bank_ring_never_recycles_a_live_slot():
    """begin_forward hands out one handle per forward; a slot whose handle is still referenced (a forward waiting for its backward)
    is never re-used -- the ring grows instead, and refuses beyond MAX_SLOTS (D steps with several penalties)."""
    import weakref
    from studiogan_amd import bank as B

    class FakeBank(B.WeightBank):
        def __init__(self, n):
            self.slots = [self._new_slot(i) for i in range(n)]
            self._ring = 0

        def _new_slot(self, s):
            sl = B._Slot()
            sl.index, sl.live = s, None
            return sl

        def take(self):
            phys = self._free_graph_slot()
            h = B._Slot()
            h.__dict__ = phys.__dict__
            phys.live = weakref.ref(h)
            return h
    fb = FakeBank(4)
    a, b, c = fb.take(), fb.take(), fb.take()
    assert [a.index, b.index, c.index] == [1, 2, 3]
    d = fb.take()                                   # all three graph slots are waiting for a backward: the ring grows
    assert d.index == 4 and len(fb.slots) == 5
    del b                                           # that forward's graph is gone
    e = fb.take()
    assert e.index == 2
    del a, c, d, e
    assert fb.take().index in (1, 3, 4)
    held = [fb.take() for _ in range(B.WeightBank.MAX_SLOTS - 1)]
    assert len(fb.slots) == B.WeightBank.MAX_SLOTS
    with pytest.raises(RuntimeError):
        fb.take()
    del held


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["biggan32", "sngan32"])
def test_boundary_through_the_references_own_loader(name):
    """The drop-in boundary driven by the REFERENCE's code: the mirrors registered as `models.<backbone>`,
    `config.define_modules` resolving its factories from studiogan_amd.ops, then the unmodified
    reference src/models/model.py:19-155 load_generator_discriminator, config.define_optimizer (src/config.py:497-563),
    utils.ema.Ema, utils.misc.make_GAN_untrainable / toggle_grad (freezeD) and strict state_dict interchange with the reference's own
    modules. Construction, .to(), deepcopy, EMA and state handling are pure host logic and run on CPU; a forward would need the GPU."""
    import importlib
    import logging
    import sys
    from oracle import ref_import as R
    from oracle import make_golden as MG
    from studiogan_amd import ops as sg_ops
    from studiogan_amd.optim import FusedAdam
    R._prepare()
    import config                                   # reference src/config.py
    y = copy.deepcopy(MG.CONFIGS[name]["yaml"])
    y["MODEL"].update(apply_g_ema=True, g_ema_decay=0.9, g_ema_start=0)
    backbone = y["MODEL"]["backbone"]
    run = {"mixed_precision": False, "distributed_data_parallel": False, "synchronized_bn": False}
    # the reference's own build of the same configuration (its utils.ops): the contract to match
    cfgs_ref = R.load_cfgs(y)
    cfgs_ref.update_cfgs(run, super="RUN")
    torch.manual_seed(0)
    Gr, Dr = R.build_models(cfgs_ref)
    saved_ops, saved_mod = config.ops, sys.modules.get("models." + backbone)
    try:
        config.ops = sg_ops                          # `import studiogan_amd.ops as ops` in src/config.py
        sys.modules["models." + backbone] = importlib.import_module("studiogan_amd.backbones." + backbone)     # §2: register under the existing name
        cfgs = R.load_cfgs(y)
        cfgs.update_cfgs(run, super="RUN")
        cfgs.define_modules()
        assert cfgs.MODULES.d_conv2d is sg_ops.snconv2d and cfgs.MODULES.g_bn is sg_ops.ConditionalBatchNorm2d
        model = importlib.import_module("models.model")
        Gen, _, _, Dis, Gen_ema, _, _, ema = model.load_generator_discriminator(
            cfgs.DATA, cfgs.OPTIMIZATION, cfgs.MODEL, cfgs.STYLEGAN, cfgs.MODULES, cfgs.RUN, "cpu", logging.getLogger("boundary-test"))
    finally:
        config.ops = saved_ops
        if saved_mod is not None:
            sys.modules["models." + backbone] = saved_mod
        else:
            sys.modules.pop("models." + backbone, None)
    assert type(Gen).__module__.startswith("studiogan_amd.") and type(Dis).__module__.startswith("studiogan_amd.")
    # EMA twin: the reference's deepcopy + its Ema class on the mirror (utils/ema.py:11-40)
    assert type(ema).__module__ == "utils.ema" and Gen_ema is not Gen
    p0, e0 = next(Gen.parameters()), next(Gen_ema.parameters())
    assert torch.equal(p0, e0) and p0.data_ptr() != e0.data_ptr()
    with torch.no_grad():
        p0.add_(1.0)
    ema.update(5)
    assert torch.allclose(e0, p0 - 1.0 + 0.1, atol=1e-6)                    # lerp(p, p_ema, 0.9)
    # same parameter / buffer names and shapes as the reference's modules; strict interchange both ways (src/utils/ckpt.py:38)
    for mine, ref in ((Gen, Gr), (Dis, Dr)):
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
        mine.load_state_dict(b, strict=True)
        ref.load_state_dict(mine.state_dict(), strict=True)
        assert [k for k, _ in mine.named_parameters()] == [k for k, _ in ref.named_parameters()]
    misc = importlib.import_module("utils.misc")
    assert misc.count_parameters(Gen) == misc.count_parameters(Gr) and misc.count_parameters(Dis) == misc.count_parameters(Dr)
    assert Gen.in_dims == Gr.in_dims and Dis.in_dims == Dr.in_dims          # misc.py:199 reads .in_dims for freezeD
    # the driver's mode / freeze helpers (misc.py:190-216,239-267,356-364)
    misc.make_GAN_untrainable(Gen, Gen_ema, Dis)
    for m in Dis.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear, nn.Embedding)):
            assert m.training, "SN layers keep iterating in eval mode (misc.py:254-262)"
    for m in Gen.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            assert not m.training
    misc.make_GAN_trainable(Gen, Gen_ema, Dis)
    misc.toggle_grad(Dis, grad=True, num_freeze_layers=1, is_stylegan=False)
    frozen = {k for k, p in Dis.named_parameters() if not p.requires_grad}
    frozen_ref = None
    misc.toggle_grad(Dr, grad=True, num_freeze_layers=1, is_stylegan=False)
    frozen_ref = {k for k, p in Dr.named_parameters() if not p.requires_grad}
    assert frozen == frozen_ref and frozen and all(k.startswith("blocks.0.") for k in frozen)
    misc.toggle_grad(Dis, grad=True, num_freeze_layers=-1, is_stylegan=False)
    # optimizer seam (src/config.py:541-563): torch.optim.Adam(eps=1e-6) as the reference builds it, and FusedAdam in its place
    cfgs.define_optimizer(Gen, Dis)
    go = cfgs.OPTIMIZATION.g_optimizer
    assert isinstance(go, torch.optim.Adam) and go.param_groups[0]["eps"] == 1e-6
    fo = FusedAdam(Gen.parameters(), lr=go.param_groups[0]["lr"], betas=go.param_groups[0]["betas"], eps=1e-6)
    fo.load_state_dict(go.state_dict())
    assert fo.param_groups[0]["lr"] == cfgs.OPTIMIZATION.g_lr
    # losses seam (src/config.py:411-433): same names, same (…, DDP=…) signature
    from studiogan_amd import losses as sg_losses
    ref_losses = importlib.import_module("utils.losses")
    for fn in ("d_hinge", "g_hinge", "d_vanilla", "g_vanilla", "d_wasserstein", "g_wasserstein", "cal_grad_penalty", "cal_deriv", "cal_r1_reg",
               "cal_maxgrad_penalty", "cal_dra_penalty", "lecam_reg", "adjust_k"):
        assert hasattr(sg_losses, fn) and hasattr(ref_losses, fn), fn
    # the product path has no CPU fallback: a forward on CPU raises instead of silently computing something else
    with pytest.raises(RuntimeError):
        Gen(torch.randn(2, cfgs.MODEL.z_dim), torch.zeros(2, dtype=torch.long))


@pytest.mark.parametrize("filt", ["bicubic", "bilinear"])
@pytest.mark.parametrize("sizes", [(32, 299), (128, 299), (512, 299), (64, 48)])
def test_pil_coefficient_tables_reproduce_pillow(filt, sizes):
    """metrics.pil_coeffs (the host half of the 'clean' / 'friendly' post-resizers, reference src/utils/resize.py:39-78) against Pillow itself:
    a float32 ('F' mode) image resized by PIL equals the separable application of the coefficient tables (horizontal pass, then vertical,
    like Pillow's ImagingResample) to float rounding. No GPU: the device kernel that consumes the tables is checked in test_eval_gpu.py."""
    import numpy as np
    from PIL import Image
    from studiogan_amd import metrics as M
    src, dst = sizes
    rng = np.random.RandomState(5)
    img = rng.rand(src, src).astype(np.float32) * 255.0
    flt = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}[filt]
    ref = np.asarray(Image.fromarray(img, mode="F").resize((dst, dst), resample=flt), dtype=np.float64)
    b, k, n = M.pil_coeffs(src, dst, filt)
    hor = np.zeros((src, dst))
    for xx in range(dst):
        x0, cnt = b[xx]
        hor[:, xx] = img[:, x0:x0 + cnt].astype(np.float64) @ k[xx, :cnt]
    out = np.zeros((dst, dst))
    for yy in range(dst):
        y0, cnt = b[yy]
        out[yy] = k[yy, :cnt] @ hor[y0:y0 + cnt]
    err = float(np.abs(out - ref).max())      # Pillow keeps the intermediate image in float32: ~1e-4 on a 0..255 range
    assert err <= 3e-3, err


def test_device_dataset_shards_like_a_distributed_sampler():
    """data.DeviceDataset under data parallelism: every rank draws the SAME per-epoch permutation and takes perm[rank::world] of it
    (reference src/loader.py:153-171: DistributedSampler(shuffle=True, drop_last=True)), so the ranks' baskets are disjoint and together cover
    world * (N // world) samples per epoch; the flip streams differ per rank; an oversized basket raises instead of coming back short.
    Index logic only (no kernel call): runs on the CPU."""
    from studiogan_amd.data import DeviceDataset
    N, world = 103, 4
    imgs = torch.zeros(N, 4, 4, 3, dtype=torch.uint8)
    labels = torch.arange(N)
    shards = [DeviceDataset(imgs, labels, device="cpu", seed=5, rank=r, world_size=world) for r in range(world)]
    per = N // world
    assert all(s.shard_len() == per for s in shards)
    seen = [torch.cat([s._next_indices(5) for _ in range(per // 5)]) for s in shards]
    allidx = torch.cat(seen)
    assert allidx.unique().numel() == allidx.numel() == world * (per // 5) * 5, "the ranks' baskets of one epoch must be disjoint"
    assert all(s.epoch == 1 for s in shards)
    # same seed, same epoch -> same permutation on every rank: rank r holds exactly perm[r::world]
    full = torch.randperm(N, generator=torch.Generator().manual_seed(5))
    for r, s in enumerate(shards):
        assert torch.equal(s._perm, full[r:per * world:world])
    # next epoch: a new shared permutation, again disjoint
    nxt = [s._next_indices(per) for s in shards]
    assert all(s.epoch == 2 for s in shards) and torch.cat(nxt).unique().numel() == per * world
    assert shards[0].flip_gen.initial_seed() != shards[1].flip_gen.initial_seed()
    with pytest.raises(RuntimeError, match="exceeds this rank's shard"):
        shards[0]._next_indices(per + 1)
    # one rank, no process group: the whole data set, as before
    ds = DeviceDataset(imgs, labels, device="cpu", seed=5)
    assert ds.world_size == 1 and ds.shard_len() == N


@pytest.mark.parametrize("relu,pool", [(False, False), (True, True)])
def test_conv_rs96_index_model(relu, pool):
    """tools/rs96_model.py replays csrc/conv_rs96.h (experimental, not yet run on a GPU) lane by lane with the kernel's own address formulas
    -- DMA piece mapping, ring slots, fragment addresses, MFMA layout, register epilogue incl. pooling -- against a direct convolution.
    Integer data: the result must be exact. Two strips per image, so strip-interior halo rows and the image borders are both walked."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("rs96_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rs96_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.run(N=1, H=4, SH=2, relu=relu, pool=pool, seed=3) == 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/configs"), reason="the reference checkout is only present in the authoring container")
def test_every_non_stylegan_reference_config_maps_to_worker_and_model_options():
    """studiogan_amd.config_map over EVERY configuration file of the reference whose backbone is not a StyleGAN one (145 files): each key / value on the
    training-step path translates into worker.Worker / backbone / ops.Modules keyword arguments that exist -- no configuration asks for something the package
    does not mirror; a handful of CIFAR10 ones are also BUILT (at width 8) through those arguments."""
    import glob
    import importlib
    import inspect
    import yaml
    import studiogan_amd  # noqa: F401
    from studiogan_amd import config_map as CM, ops
    from studiogan_amd.worker import Worker
    sig = set(inspect.signature(Worker.__init__).parameters)
    files = sorted(glob.glob("/root/reference/src/configs/*/*.yaml"))
    seen, flags = 0, set()
    for f in files:
        y = yaml.safe_load(open(f))
        if "stylegan" in (y.get("MODEL") or {}).get("backbone", "resnet"):
            with pytest.raises(NotImplementedError):
                CM.model_args(y)
            continue
        kw = CM.worker_kwargs(y)
        assert not (set(kw) - sig), (f, set(kw) - sig)
        bb, mods, gen, dis = CM.model_args(y)
        mod = importlib.import_module("studiogan_amd.backbones." + bb)
        assert not (set(gen) - set(inspect.signature(mod.Generator.__init__).parameters)), f
        assert not (set(dis) - set(inspect.signature(mod.Discriminator.__init__).parameters)), f
        assert not (set(mods) - set(inspect.signature(ops.Modules.__init__).parameters)), f
        flags |= {k for k, v in kw.items() if k.startswith("apply_") and v} | ({"info"} if kw["info_type"] != "N/A" else set())
        seen += 1
    assert seen >= 140
    assert {"apply_diffaug", "apply_ada", "apply_apa", "apply_cr", "apply_bcr", "apply_zcr", "apply_gp", "apply_dra", "apply_maxgp", "apply_r1_reg", "apply_wc", "apply_fm",
            "apply_lo", "apply_lecam", "apply_g_ema", "info"} <= flags
    for name in ("BigGAN-Info", "SNGAN-ADA", "LOGAN", "DCGAN-Info", "ReACGAN-ADC-DiffAug", "MHGAN", "WGAN-WC", "BigGAN-Deep"):
        y = yaml.safe_load(open(f"/root/reference/src/configs/CIFAR10/{name}.yaml"))
        y.setdefault("MODEL", {})
        for k in ("g_conv_dim", "d_conv_dim"):
            if y["MODEL"].get(k, 64) != "N/A":
                y["MODEL"][k] = 8
        bb, mods, gen, dis = CM.model_args(y)
        mod = importlib.import_module("studiogan_amd.backbones." + bb)
        MOD, MODEL = ops.Modules(**mods), CM.model_namespace(y)
        G = mod.Generator(mixed_precision=False, MODULES=MOD, MODEL=MODEL, **gen)
        D = mod.Discriminator(mixed_precision=False, MODULES=MOD, MODEL=MODEL, **dis)
        assert sum(p.numel() for p in G.parameters()) > 0 and sum(p.numel() for p in D.parameters()) > 0, name
        if CM.worker_kwargs(y)["info_type"] != "N/A":
            assert hasattr(D, "info_discrete_linear") and (hasattr(G, "info_proj_linear") or hasattr(G, "info_mix_linear")), name


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
@pytest.mark.parametrize("prior,trunc,info", [("gaussian", -1.0, "N/A"), ("gaussian", 0.7, "N/A"), ("uniform", -1.0, "N/A"), ("gaussian", -1.0, "both"),
                                              ("gaussian", -1.0, "discrete"), ("uniform", -1.0, "continuous")])
def test_evaluation_time_sampling_consumes_the_generators_like_the_reference(prior, trunc, info):
    """worker.sample_latents == what the reference's sample.generate_images(is_train=False) hands the generator (src/utils/sample.py:90-118,160): same labels, latents
    (device normal / scipy's truncated normal / host uniform) and InfoGAN codes behind z under the same torch + numpy seeds -- pure host logic, runs on the CPU."""
    import importlib
    import types
    import numpy as np
    from oracle import ref_import as R
    from studiogan_amd.worker import sample_latents
    R._prepare()
    sample = importlib.import_module("utils.sample")
    MODEL = types.SimpleNamespace(info_type=info, info_num_discrete_c=3, info_dim_discrete_c=5, info_num_conti_c=2, backbone="resnet")
    LOSS = types.SimpleNamespace(apply_lo=False, lo_steps4train=2, lo_steps4eval=2)
    RUN = types.SimpleNamespace(langevin_sampling=False)
    got = {}

    def gen(zs, ys, eval):
        got["zs"], got["ys"], got["eval"] = zs.clone(), ys.clone(), eval
        return zs
    torch.manual_seed(11)
    np.random.seed(12)
    sample.generate_images(z_prior=prior, truncation_factor=trunc, batch_size=6, z_dim=16, num_classes=10, y_sampler="totally_random", radius="N/A", generator=gen,
                           discriminator=None, is_train=False, LOSS=LOSS, RUN=RUN, MODEL=MODEL, device="cpu", is_stylegan=False, generator_mapping=None,
                           generator_synthesis=None, style_mixing_p=0.0, stylegan_update_emas=False, cal_trsp_cost=False)
    torch.manual_seed(11)
    np.random.seed(12)
    zs, ys = sample_latents(6, 16, 10, torch.device("cpu"), z_prior=prior, truncation_factor=trunc, MODEL=MODEL)
    assert got["eval"] is True
    assert torch.equal(ys, got["ys"]) and zs.shape == got["zs"].shape and torch.equal(zs.float(), got["zs"].float())
    if trunc > 0:
        assert float(zs.abs().max()) <= trunc


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/configs"), reason="the reference checkout is only present in the authoring container")
def test_seeded_construction_reproduces_the_references_initialisation():
    """config_map.build under torch.manual_seed(s) == the reference's Generator / Discriminator built by ITS code under the same seed, bit for bit (every layer is created,
    and its initial weights and spectral-norm vectors drawn, in the reference's order): all 55 non-StyleGAN CIFAR10 configuration files at width 8. Host logic only (the
    construction runs on the CPU; nothing is launched). tests/aug_checks.py config_step_case relies on it: its fixture stores no weights."""
    import glob
    import yaml
    from oracle import ref_import as R
    from studiogan_amd import config_map as CM
    n = 0
    for f in sorted(glob.glob("/root/reference/src/configs/CIFAR10/*.yaml")):
        y = yaml.safe_load(open(f))
        if "stylegan" in (y.get("MODEL") or {}).get("backbone", "resnet"):
            continue
        M = y.setdefault("MODEL", {})
        for k in ("g_conv_dim", "d_conv_dim"):
            if M.get(k, 64) != "N/A":
                M[k] = 8
        for k in ("d_embed_dim", "g_shared_dim"):
            if M.get(k, "N/A") != "N/A":
                M[k] = 16
        if M.get("backbone") == "deep_conv" and n > 0 and "Info" not in f:
            continue          # (full-width DCGAN: one plain file and the InfoGAN one are enough)
        torch.manual_seed(5)
        cfgs = R.load_cfgs({k: v for k, v in y.items() if k in ("DATA", "MODEL", "LOSS", "OPTIMIZATION", "AUG")})
        Gr, Dr = R.build_models(cfgs)
        torch.manual_seed(5)
        G, D, _ = CM.build(y, torch.device("cpu"))
        sg, sd = G.state_dict(), D.state_dict()
        assert all(torch.equal(sg[k], v) for k, v in Gr.state_dict().items()), os.path.basename(f)
        assert all(torch.equal(sd[k], v) for k, v in Dr.state_dict().items()), os.path.basename(f)
        n += 1
    assert n >= 50


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_reference_worker_drives_amd_backbones(tmp_path):
    """The seam crossed with the REFERENCE's driver on top: `src/models/big_resnet_amd.py` is the three-line file a maintainer adds (written
    here into a second portion of the reference's `models` namespace package), `config.ops` / `config.losses` are this package's modules, and then nothing but
    reference code runs: `Configurations` (define_modules, define_losses, define_optimizer), `models.model.load_generator_discriminator`
    (src/models/model.py:19-22 `__import__("models." + backbone)`), `utils.ema.Ema`, `WORKER.__init__`, and the unmodified `WORKER.train_discriminator` /
    `WORKER.train_generator` (src/worker.py:213-681: sample.generate_images, misc.toggle_grad / make_GAN_trainable / untrack_bn_statistics, torch.optim.Adam, Ema.update)
    for ImageNet/BigGAN-256.yaml (C3: 128 x 128, cBN + PD + attention, hinge, EMA, two discriminator updates) at width 8, batch 4. The kernels execute on the CPU
    interpreter (tests/hipemu). Compared with the same driver over the reference's own networks from the same initial state, seed and baskets: losses, every
    gradient of the last discriminator update and of the generator update, parameters after Adam (in units of lr), the EMA twin."""
    import importlib
    import logging
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import emu
    if not emu.available():
        pytest.skip("host clang++ of the ROCm toolchain not found")
    import fullemu
    from oracle import ref_import as R
    from studiogan_amd import ops as sg_ops, losses as sg_losses
    from config_worker_parity_emulated import reference_worker, grads, params, worst, N_D
    from config_parity_emulated import shrink
    R._prepare()
    import config
    y = shrink(yaml.safe_load(open("/root/reference/src/configs/ImageNet/BigGAN-256.yaml")))
    batch = 4
    y["OPTIMIZATION"].update(batch_size=batch, d_updates_per_step=N_D)
    y["MODEL"].update(g_ema_start=0, g_ema_decay=0.9)          # (the file starts the average after 20000 steps: here it is live at step 1)
    S, nc = y["DATA"]["img_size"], y["DATA"]["num_classes"]
    g = torch.Generator().manual_seed(11)
    baskets = [(torch.randint(0, 256, (N_D * batch, 3, S, S), generator=g).float() / 127.5 - 1.0, torch.randint(0, nc, (N_D * batch,), generator=g))]
    nt = torch.get_num_threads()

    def drive(cfgs, Gen, Dis):
        w, Gema = reference_worker(R, cfgs, Gen, Dis, baskets, "N/A")
        torch.manual_seed(77)
        _, d_loss = w.train_discriminator(1)
        dg = grads(Dis)
        g_loss = w.train_generator(1)
        return dict(d_loss=float(d_loss.detach()), g_loss=float(g_loss.detach()), dg=dg, gg=grads(Gen), dp=params(Dis), gp=params(Gen), ema=params(Gema))

    # ---- the reference's driver over the reference's networks
    torch.manual_seed(0)
    cfgs_r = R.load_cfgs(y)
    Gr, Dr = R.build_models(cfgs_r)
    g_state, d_state = copy.deepcopy(Gr.state_dict()), copy.deepcopy(Dr.state_dict())
    ref = drive(cfgs_r, Gr, Dr)
    # ---- the same driver over this package's networks, found through the reference's own import
    (tmp_path / "models").mkdir()
    (tmp_path / "models" / "big_resnet_amd.py").write_text(
        f"import sys; sys.path.insert(0, {ROOT!r})\n"
        "import studiogan_amd\n"
        "from studiogan_amd.backbones.big_resnet import Generator, Discriminator\n")
    saved = (config.ops, config.losses)
    sys.path.insert(0, str(tmp_path))
    try:
        config.ops, config.losses = sg_ops, sg_losses           # the two import lines of src/config.py
        y2 = copy.deepcopy(y)
        y2["MODEL"]["backbone"] = "big_resnet_amd"
        cfgs = R.load_cfgs(y2)
        cfgs.update_cfgs({"mixed_precision": False, "distributed_data_parallel": False, "synchronized_bn": False}, super="RUN")      # (what src/main.py's flags set)
        assert cfgs.MODULES.d_conv2d is sg_ops.snconv2d and cfgs.MODULES.g_bn is sg_ops.ConditionalBatchNorm2d
        model = importlib.import_module("models.model")
        torch.set_num_threads(1)
        with fullemu.Installed(dma_late=1, greedy=1, seed=1) as E:
            Gen, _, _, Dis, _, _, _, _ = model.load_generator_discriminator(cfgs.DATA, cfgs.OPTIMIZATION, cfgs.MODEL, cfgs.STYLEGAN, cfgs.MODULES, cfgs.RUN, "cpu",
                                                                            logging.getLogger("seam-test"))
            assert type(Gen).__module__ == "studiogan_amd.backbones.big_resnet" and sys.modules["models.big_resnet_amd"].Generator is type(Gen)
            Gen.load_state_dict(g_state, strict=True)
            Dis.load_state_dict(d_state, strict=True)
            launches0 = E.counters()["launches"]
            mine = drive(cfgs, Gen, Dis)
            assert cfgs.LOSS.d_loss is sg_losses.d_hinge and isinstance(cfgs.OPTIMIZATION.d_optimizer, torch.optim.Adam)
            assert E.counters()["launches"] - launches0 > 500, "the update did not run through the kernels"
    finally:
        torch.set_num_threads(nt)
        config.ops, config.losses = saved
        sys.path.remove(str(tmp_path))
        sys.modules.pop("models.big_resnet_amd", None)
    lr_d, lr_g = y["OPTIMIZATION"]["d_lr"], y["OPTIMIZATION"]["g_lr"]
    e = dict(d_loss=abs(mine["d_loss"] - ref["d_loss"]) / max(abs(ref["d_loss"]), 1e-3), g_loss=abs(mine["g_loss"] - ref["g_loss"]) / max(abs(ref["g_loss"]), 1e-3),
             d_grads=worst(mine["dg"], ref["dg"])[0], g_grads=worst(mine["gg"], ref["gg"])[0],
             d_params_lr=max(float((mine["dp"][k] - ref["dp"][k]).abs().max()) for k in ref["dp"]) / (N_D * lr_d),
             g_params_lr=max(float((mine["gp"][k] - ref["gp"][k]).abs().max()) for k in ref["gp"]) / lr_g,
             ema_lr=max(float((mine["ema"][k] - ref["ema"][k]).abs().max()) for k in ref["ema"]) / lr_g)
    print({k: f"{v:.2e}" for k, v in e.items()})
    # (fp32 both sides; bounds of tools/config_worker_parity_emulated.py: an element whose gradient is rounding noise moves by about +-lr per Adam step either way)
    assert e["d_loss"] <= 2e-3 and e["g_loss"] <= 2e-3 and e["d_grads"] <= 1e-2 and e["g_grads"] <= 1e-2, e
    assert e["d_params_lr"] <= 3.0 and e["g_params_lr"] <= 3.0 and e["ema_lr"] <= 3.0, e
