"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference with stubs) on CPU and
check the restatement (oracle/restate.py) against it on the same inputs.

    python oracle/make_golden.py            # writes fixtures, prints max |restate - reference| per tensor family

Runs only in the authoring container (needs /root/reference). The fixtures are committed; the GPU box never needs the
reference. TEST INFRASTRUCTURE ONLY.

A fixture holds, for one small configuration: the initial G/D parameters + buffers, the synthetic inputs (z, labels,
uint8-grid real images; all from a seeded CPU generator as SURVEY.md §8d prescribes), and what the reference produced
over one training step (n_d discriminator updates + one generator update, reference src/loader.py:392-405):
fake images, logits, losses, every parameter gradient of the first D update and of the G update, and the final
parameters / buffers (spectral-norm u,v; BN running statistics) after the Adam steps.
"""
import importlib
import json
import math
import os
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import as R  # noqa: E402
from oracle import restate as O  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CONFIGS = {
    # small BigGAN (same code path as C3: cBN + PD + SN on both + attention in G and D, hinge)
    "biggan32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "big_resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                        "apply_attn": True, "attn_g_loc": [2], "attn_d_loc": [1], "z_dim": 40, "g_shared_dim": 16, "g_conv_dim": 8, "d_conv_dim": 8},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.00005, "d_lr": 0.0002, "beta1": 0.0, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=1234),
    # BigGAN-deep (C4 family: configs/ImageNet/BigGAN-Deep-256.yaml at width 8): bottleneck blocks, depth 2, attention, SN both
    "bigdeep32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "big_resnet_deep_legacy", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                        "apply_attn": True, "attn_g_loc": [2], "attn_d_loc": [1], "z_dim": 24, "g_shared_dim": 16, "g_conv_dim": 8, "d_conv_dim": 8,
                        "g_depth": 2, "d_depth": 2},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.00005, "d_lr": 0.0002, "beta1": 0.0, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=4242),
    # StudioGAN's own BigGAN-deep variant (models/big_resnet_deep_studiogan.py): learned 1x1 skips, optblock, narrow 32x32 stem
    "bigdeepsg32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "big_resnet_deep_studiogan", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                        "apply_attn": True, "attn_g_loc": [2], "attn_d_loc": [1], "z_dim": 24, "g_shared_dim": 16, "g_conv_dim": 8, "d_conv_dim": 8,
                        "g_depth": 2, "d_depth": 2},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.00005, "d_lr": 0.0002, "beta1": 0.0, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=2424),
    # SNGAN on the ResNet backbone (C2 family: reference configs/CIFAR10/SNGAN.yaml at width 8): cBN on the one-hot label, PD, SN in D only
    "sngan32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.0002, "d_lr": 0.0002, "beta1": 0.5, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=4321),
    # unconditional ResNet GAN with batch norm in D (no SN anywhere; the C5 WGAN-GP model family with the vanilla loss)
    "resgan32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "resnet", "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8},
              "LOSS": {"adv_loss": "vanilla"},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.0002, "d_lr": 0.0002, "beta1": 0.5, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=777),
    # WGAN-GP (C5 family: configs/CIFAR10/WGAN-GP.yaml at width 8): wasserstein + gradient penalty (double backward through a
    # discriminator with batch norm), unconditional, no spectral norm
    "wgangp32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "resnet", "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8},
              "LOSS": {"adv_loss": "wasserstein", "apply_gp": True, "gp_lambda": 10.0},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.0002, "d_lr": 0.0002, "beta1": 0.5, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=555),
    # the same penalty through a spectrally normalised projection discriminator with self-attention-free ResNet blocks
    "sngp32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8},
              "LOSS": {"adv_loss": "hinge", "apply_gp": True, "gp_lambda": 10.0},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.0002, "d_lr": 0.0002, "beta1": 0.5, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=556),
    # DCGAN exactly as configs/CIFAR10/DCGAN.yaml (C1): widths are hard-coded in models/deep_conv.py (6.7 M parameters),
    # so the fixture is COMPACT: formula-generated initial state + samples/norms of the large expected tensors
    "dcgan32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "deep_conv", "g_conv_dim": "N/A", "d_conv_dim": "N/A"},
              "OPTIMIZATION": {"batch_size": 8, "d_updates_per_step": 2}},
        batch=8, n_d=2, seed=99, compact=True),
    # DCGAN widths with spectral norm on G (ConvTranspose2d, dim=1) and D, cBN + PD, hinge
    "sndcgan32": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "deep_conv", "g_conv_dim": "N/A", "d_conv_dim": "N/A", "g_cond_mtd": "cBN", "d_cond_mtd": "PD",
                        "apply_g_sn": True, "apply_d_sn": True},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 8, "d_updates_per_step": 2}},
        batch=8, n_d=2, seed=100, compact=True),
    # ---- FULL-WIDTH fixtures: the configurations bench.py measures, at their real channel widths / resolutions, so that the kernels the
    # benchmark dispatches (halo conv_v3, streaming conv_sk, conv_v2, wgrad_v2, fused attention at HW = 4096) are the ones under test.
    # C3 = configs/ImageNet/BigGAN-256.yaml verbatim (ch 96, 128^2, z 120, shared 128, 1000 classes, attention G@64^2 / D@64^2), batch 4
    "biggan128w": dict(
        yaml={"DATA": {"name": "ImageNet", "img_size": 128, "num_classes": 1000},
              "MODEL": {"backbone": "big_resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                        "apply_attn": True, "attn_g_loc": [4], "attn_d_loc": [1], "z_dim": 120, "g_shared_dim": 128, "g_conv_dim": 96, "d_conv_dim": 96},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.00005, "d_lr": 0.0002, "beta1": 0.0, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=4, n_d=2, seed=31, compact=True, sample=512),
    # C2 = configs/CIFAR10/SNGAN.yaml verbatim (resnet, ch 64: G 256 channels, D 128), batch 16
    "sngan32w": dict(
        yaml={"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 16, "d_updates_per_step": 2}},
        batch=16, n_d=2, seed=32, compact=True, sample=512),
    # C5 = configs/CIFAR10/WGAN-GP.yaml at img_size 128 (resnet ch 64, unconditional, no SN => batch norm in D, wasserstein + GP), batch 2
    "wgangp128w": dict(
        yaml={"DATA": {"name": "ImageNet", "img_size": 128, "num_classes": 1000},
              "MODEL": {"backbone": "resnet"},
              "LOSS": {"adv_loss": "wasserstein", "apply_gp": True, "gp_lambda": 10.0},
              "OPTIMIZATION": {"batch_size": 2, "d_updates_per_step": 2}},
        batch=2, n_d=2, seed=33, compact=True, sample=512),
    # C4 family = configs/ImageNet/BigGAN-Deep-2048.yaml model section verbatim (big_resnet_deep_legacy, ch 128, depth 2, 128^2), batch 2
    "bigdeep128w": dict(
        yaml={"DATA": {"name": "ImageNet", "img_size": 128, "num_classes": 1000},
              "MODEL": {"backbone": "big_resnet_deep_legacy", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                        "apply_attn": True, "attn_g_loc": [4], "attn_d_loc": [1], "z_dim": 128, "g_shared_dim": 128, "g_conv_dim": 128, "d_conv_dim": 128,
                        "g_depth": 2, "d_depth": 2},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 2, "g_lr": 0.00005, "d_lr": 0.0002, "beta1": 0.0, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=2, n_d=2, seed=34, compact=True, sample=512),
    # C4 at the resolution BASELINE.json names ("BigGAN-Deep ImageNet-256"): the same yaml with img_size 256 -- the model tables support it
    # (reference src/models/big_resnet_deep_legacy.py:84,240; SURVEY.md 8(d) "report both 128^2 and 256^2"); attention G@64^2, D@128^2 (HW = 16384)
    "bigdeep256w": dict(
        yaml={"DATA": {"name": "ImageNet", "img_size": 256, "num_classes": 1000},
              "MODEL": {"backbone": "big_resnet_deep_legacy", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                        "apply_attn": True, "attn_g_loc": [4], "attn_d_loc": [1], "z_dim": 128, "g_shared_dim": 128, "g_conv_dim": 128, "d_conv_dim": 128,
                        "g_depth": 2, "d_depth": 2},
              "LOSS": {"adv_loss": "hinge"},
              "OPTIMIZATION": {"batch_size": 2, "g_lr": 0.00005, "d_lr": 0.0002, "beta1": 0.0, "beta2": 0.999, "d_updates_per_step": 2}},
        batch=2, n_d=2, seed=35, compact=True, sample=512),
}

SAMPLE = 2048          # compact fixtures: tensors above FULL_MAX elements keep SAMPLE evenly spaced values + [sum, l2]
FULL_MAX = 8192


def sample_index(numel, n=SAMPLE):
    return (torch.arange(n, dtype=torch.int64) * numel) // n


def formula_state(spec, seed):
    """Deterministic initial state for compact fixtures: numpy RandomState (bit-stable by specification) keyed by the
    tensor name. spec: {name: shape}. Weights ~ N(0, 1/fan_in), BN gains 1 + 0.1 N, biases 0.1 N, unit u / v."""
    out = {}
    for name, shape in spec.items():
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
        leaf = name.rsplit(".", 1)[-1]
        shape = tuple(shape)
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_mean":
            t = torch.zeros(shape)
        elif leaf == "running_var":
            t = torch.ones(shape)
        else:
            v = rs.standard_normal(shape).astype(np.float64)
            if leaf in ("weight_u", "weight_v"):
                v = v / np.linalg.norm(v)
            elif len(shape) >= 2:
                v = v / math.sqrt(int(np.prod(shape[1:])))
            elif leaf == "weight":
                v = 1.0 + 0.1 * v
            else:
                v = 0.1 * v
            t = torch.from_numpy(v.astype(np.float32))
        out[name] = t
    return out


def compact_entries(key, v, sample=SAMPLE):
    """fixture entries for one expected tensor of a compact fixture."""
    if v.numel() <= min(FULL_MAX, 4 * sample):
        return {"exp/" + key: v}
    flat = v.reshape(-1)
    return {"exps/" + key: flat[sample_index(flat.numel(), sample)].clone(),
            "expn/" + key: torch.tensor([float(flat.double().sum()), float(flat.double().norm())], dtype=torch.float64)}


def oracle_cfg(y):
    M, D = y["MODEL"], y["DATA"]
    return dict(img_size=D["img_size"], num_classes=D["num_classes"], g_conv_dim=M.get("g_conv_dim", 64), d_conv_dim=M.get("d_conv_dim", 64),
                z_dim=M.get("z_dim", 128), attn_g_loc=M.get("attn_g_loc", []), attn_d_loc=M.get("attn_d_loc", []), apply_attn=M.get("apply_attn", False),
                g_cond_mtd=M.get("g_cond_mtd", "W/O"), d_cond_mtd=M.get("d_cond_mtd", "W/O"), apply_g_sn=M.get("apply_g_sn", False),
                apply_d_sn=M.get("apply_d_sn", False), backbone=M.get("backbone", "resnet"), g_shared_dim=M.get("g_shared_dim", 0),
                g_depth=M.get("g_depth", 1), d_depth=M.get("d_depth", 1), aux_cls_type=M.get("aux_cls_type", "W/O"),
                normalize_d_embed=M.get("normalize_d_embed", False), d_embed_dim=M.get("d_embed_dim", "N/A"))


def synth_inputs(seed, n_d, batch, z_dim, num_classes, img_size):
    g = torch.Generator().manual_seed(seed)
    ins = {}
    for i in range(n_d + 1):
        ins[f"z{i}"] = torch.randn(batch, z_dim, generator=g)
        ins[f"fl{i}"] = torch.randint(0, num_classes, (batch,), generator=g)
    for i in range(n_d):
        k = torch.randint(0, 256, (batch, 3, img_size, img_size), generator=g)
        ins[f"real{i}"] = k.float() / 127.5 - 1.0
        ins[f"rl{i}"] = torch.randint(0, num_classes, (batch,), generator=g)
    return ins


GP_SEED = 100000   # host-RNG seed (+ update index) in front of each gradient-penalty alpha draw (utils/losses.py:303)


def gp_alpha(seed, i, batch):
    torch.manual_seed(seed + GP_SEED + i)
    return torch.rand(batch, 1)


def run_reference(cfgs, Gen, Dis, ins, n_d, seed=0):
    misc = importlib.import_module("utils.misc")
    ref_losses = importlib.import_module("utils.losses")
    cfgs.define_losses()
    cfgs.define_optimizer(Gen, Dis)
    g_opt, d_opt = cfgs.OPTIMIZATION.g_optimizer, cfgs.OPTIMIZATION.d_optimizer
    exp = {}
    for i in range(n_d):  # worker.train_discriminator (src/worker.py:213-497)
        misc.make_GAN_trainable(Gen, None, Dis)
        misc.toggle_grad(Gen, False)
        misc.toggle_grad(Dis, True)
        Gen.apply(misc.untrack_bn_statistics)
        d_opt.zero_grad()
        fake = Gen(ins[f"z{i}"], ins[f"fl{i}"])
        rd = Dis(ins[f"real{i}"], ins[f"rl{i}"])
        fd = Dis(fake, ins[f"fl{i}"])
        loss = cfgs.LOSS.d_loss(rd["adv_output"], fd["adv_output"], DDP=False)
        if cfgs.LOSS.apply_gp:   # src/worker.py:369-375
            torch.manual_seed(seed + GP_SEED + i)
            gp = ref_losses.cal_grad_penalty(real_images=ins[f"real{i}"], real_labels=ins[f"rl{i}"], fake_images=fake, discriminator=Dis, device="cpu")
            loss = loss + cfgs.LOSS.gp_lambda * gp
            if i == 0:
                exp["gp0"] = gp.detach().clone()
        loss.backward()
        if i == 0:
            exp["fake0"], exp["adv_r0"], exp["adv_f0"] = fake.detach().clone(), rd["adv_output"].detach().clone(), fd["adv_output"].detach().clone()
            exp["d_loss0"] = loss.detach().clone()
            for k, p in Dis.named_parameters():
                exp["D_grad0/" + k] = p.grad.detach().clone()
        d_opt.step()
    misc.make_GAN_trainable(Gen, None, Dis)  # worker.train_generator (src/worker.py:502-681)
    misc.toggle_grad(Dis, False)
    misc.toggle_grad(Gen, True)
    Gen.apply(misc.track_bn_statistics)
    g_opt.zero_grad()
    fake = Gen(ins[f"z{n_d}"], ins[f"fl{n_d}"])
    fd = Dis(fake, ins[f"fl{n_d}"])
    loss = cfgs.LOSS.g_loss(fd["adv_output"], DDP=False)
    loss.backward()
    exp["g_loss"] = loss.detach().clone()
    exp["fake_g"] = fake.detach().clone()
    for k, p in Gen.named_parameters():
        exp["G_grad/" + k] = p.grad.detach().clone()
    g_opt.step()
    for k, v in list(Gen.named_parameters()) + list(Gen.named_buffers()):
        exp["G_final/" + k] = v.detach().clone()
    for k, v in list(Dis.named_parameters()) + list(Dis.named_buffers()):
        exp["D_final/" + k] = v.detach().clone()
    return exp


def run_restatement(ocfg, y, GP, GB, DP, DB, ins, n_d, seed=0):
    """Same step through oracle/restate.py (this is also what the GPU tests execute as the checker)."""
    opt = y.get("OPTIMIZATION", {})
    g_lr, d_lr = opt.get("g_lr", 0.0002), opt.get("d_lr", 0.0002)
    b1, b2 = opt.get("beta1", 0.5), opt.get("beta2", 0.999)
    kind = y.get("LOSS", {}).get("adv_loss", "vanilla")
    lam = y.get("LOSS", {}).get("gp_lambda", 10.0) if y.get("LOSS", {}).get("apply_gp", False) else None
    gen_fn, dis_fn = O.model_fns(ocfg)
    g_opt, d_opt = O.AdamState(GP, g_lr, b1, b2), O.AdamState(DP, d_lr, b1, b2)
    exp = {}
    for i in range(n_d):
        out = O.d_update(gen_fn, dis_fn, GP, GB, DP, DB, d_opt, [ins[f"real{i}"]], [ins[f"rl{i}"]], [ins[f"z{i}"]], [ins[f"fl{i}"]], kind,
                         record=(i == 0), gp_lambda=lam, gp_alpha=[gp_alpha(seed, i, ins[f"z{i}"].shape[0])] if lam is not None else None)
        if i == 0 and lam is not None:
            exp["gp0"] = out["gp"]
        if i == 0:
            exp["fake0"], exp["adv_r0"], exp["adv_f0"] = out["fake"], out["adv_r"], out["adv_f"]
            exp["d_loss0"] = torch.tensor(out["loss"])
            for k, g in out["grads"].items():
                exp["D_grad0/" + k] = g
    out = O.g_update(gen_fn, dis_fn, GP, GB, DP, DB, g_opt, [ins[f"z{n_d}"]], [ins[f"fl{n_d}"]], kind, record=True)
    exp["g_loss"] = torch.tensor(out["loss"])
    exp["fake_g"] = out["fake"]
    for k, g in out["grads"].items():
        exp["G_grad/" + k] = g
    for k, v in list(GP.items()) + list(GB.items()):
        exp["G_final/" + k] = v.detach().clone()
    for k, v in list(DP.items()) + list(DB.items()):
        exp["D_final/" + k] = v.detach().clone()
    return exp


COND_EPS = 2e-6     # relative size of the weight perturbation of the conditioning measurement (fp32 kernels of two implementations
                    # disagree at about this level: first-forward quantities of the HIP path match the oracle to ~5e-6)


def _perturbed(P, eps, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: (v * (1 + eps * torch.randn(v.shape, generator=g)) if v.is_floating_point() else v.clone()) for k, v in P.items()}


def _noise(a, b):
    d = (a.detach().double() - b.detach().double()).reshape(-1)
    return torch.tensor([float(d.norm() / max(d.numel(), 1) ** 0.5), float(d.abs().max()) if d.numel() else 0.0], dtype=torch.float64)


def conditioning(name):
    """How far the ORACLE ITSELF moves when every weight is perturbed by COND_EPS (relative): per expected tensor [rms, max] of the
    difference, (a) over the unsynchronised chain of the golden step ("chain/<key>", keys of run_restatement) and (b) per update with
    the perturbation applied right before that update ("stage/D<i>/..", "stage/G/..": what the re-synchronised stage-wise test
    sees). A WGAN-GP critic at batch 2, or a generator gradient that has passed every ReLU of D after two Adam updates, turn such a
    perturbation into 1e-3..1e-1 relative differences; the GPU tests allow base tolerance + 4 x this measured noise instead of a
    hand-picked number. Written to tests/golden/<name>.cond.npz; needs only the restatement (no reference)."""
    import copy
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN_DIR)))
    from util import load_golden, sub, hyper
    fix, meta = load_golden(name)
    y, n_d, seed = meta["yaml"], meta["n_d"], meta["seed"]
    ocfg = oracle_cfg(y)
    isb = lambda k: any(s in k for s in ("weight_u", "weight_v", "running_", "num_batches"))
    GI, DI = sub(fix, "G_init/"), sub(fix, "D_init/")
    split = lambda d: ({k: v.clone() for k, v in d.items() if not isb(k)}, {k: v.clone() for k, v in d.items() if isb(k)})
    ins = sub(fix, "in/")
    out = {}
    # (a) chain
    GP, GB = split(GI); DP, DB = split(DI)
    base = run_restatement(ocfg, y, GP, GB, DP, DB, ins, n_d, seed)
    GP, GB = split(GI); DP, DB = split(DI)
    pert = run_restatement(ocfg, y, _perturbed(GP, COND_EPS, 11), GB, _perturbed(DP, COND_EPS, 12), DB, ins, n_d, seed)
    for k in base:
        out["chain/" + k] = _noise(base[k], pert[k])
    # (b) per update, perturbation applied to the state the update starts from
    opt = hyper(y)
    kind = opt["adv_loss"]
    lam = opt["gp_lambda"] if opt["apply_gp"] else None
    gen_fn, dis_fn = O.model_fns(ocfg)
    GP, GB = split(GI); DP, DB = split(DI)
    g_opt, d_opt = O.AdamState(GP, opt["g_lr"], opt["beta1"], opt["beta2"]), O.AdamState(DP, opt["d_lr"], opt["beta1"], opt["beta2"])
    B = ins["z0"].shape[0]
    for i in range(n_d + 1):
        st = copy.deepcopy((GP, GB, DP, DB, g_opt, d_opt))
        pGP, pGB, pDP, pDB, pg, pd = copy.deepcopy(st)
        pGP, pDP = _perturbed(pGP, COND_EPS, 21 + i), _perturbed(pDP, COND_EPS, 31 + i)
        res = []
        for (a, b, c, d, go, do) in ((GP, GB, DP, DB, g_opt, d_opt), (pGP, pGB, pDP, pDB, pg, pd)):
            if i < n_d:
                o = O.d_update(gen_fn, dis_fn, a, b, c, d, do, [ins[f"real{i}"]], [ins[f"rl{i}"]], [ins[f"z{i}"]], [ins[f"fl{i}"]], kind, record=True,
                               gp_lambda=lam, gp_alpha=[gp_alpha(seed, i, B)] if lam is not None else None)
                res.append((o, c))
            else:
                o = O.g_update(gen_fn, dis_fn, a, b, c, d, go, [ins[f"z{n_d}"]], [ins[f"fl{n_d}"]], kind, record=True)
                res.append((o, a))
        tag = f"stage/D{i}/" if i < n_d else "stage/G/"
        (o0, p0), (o1, p1) = res
        for k in ("fake", "adv_r", "adv_f", "gp"):
            if k in o0 and torch.is_tensor(o0[k]):
                out[tag + k] = _noise(o0[k], o1[k])
        for k in o0["grads"]:
            if o0["grads"][k] is not None:
                out[tag + "grad/" + k] = _noise(o0["grads"][k], o1["grads"][k])
        for k in p0:
            out[tag + "param/" + k] = _noise(p0[k], p1[k])     # (the 2e-6 start offset is far below the +-lr movement of an Adam step)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".cond.npz"), **{k: v.numpy() for k, v in out.items()})
    worst = {}
    for k, v in out.items():
        fam = "/".join(k.split("/")[:3]) if k.startswith("stage") else k.split("/")[1]
        worst[fam] = max(worst.get(fam, 0.0), float(v[0]))
    print(name, "conditioning (rms of the oracle's own movement under a", COND_EPS, "relative weight perturbation), worst per family:")
    for fam, e in sorted(worst.items()):
        print(f"   {fam:28s} {e:.3e}")


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "--cond":
        for name in sys.argv[2:]:
            conditioning(name)
        return
    assert R.available(), "/root/reference is required to (re)generate the golden fixtures"
    only = sys.argv[1:]
    for name, c in CONFIGS.items():
        if only and name not in only:
            continue
        y = c["yaml"]
        cfgs = R.load_cfgs(y)
        torch.manual_seed(c["seed"])
        Gen, Dis = R.build_models(cfgs)
        compact = c.get("compact", False)
        meta = {"yaml": y, "batch": c["batch"], "n_d": c["n_d"], "seed": c["seed"]}
        if compact:
            meta["compact"] = True
            if c.get("sample"):
                meta["sample"] = c["sample"]
            meta["G_spec"] = {k: list(v.shape) for k, v in Gen.state_dict().items()}
            meta["D_spec"] = {k: list(v.shape) for k, v in Dis.state_dict().items()}
            Gen.load_state_dict(formula_state(meta["G_spec"], c["seed"]), strict=True)
            Dis.load_state_dict(formula_state(meta["D_spec"], c["seed"] + 1), strict=True)
        GP, GB = R.split_state(Gen)
        DP, DB = R.split_state(Dis)
        ocfg = oracle_cfg(y)
        ins = synth_inputs(c["seed"] + 1, c["n_d"], c["batch"], ocfg["z_dim"], ocfg["num_classes"], ocfg["img_size"])
        fix = {}
        if not compact:
            for k, v in list(GP.items()) + list(GB.items()):
                fix["G_init/" + k] = v.clone()
            for k, v in list(DP.items()) + list(DB.items()):
                fix["D_init/" + k] = v.clone()
        for k, v in ins.items():
            fix["in/" + k] = v
        exp_ref = run_reference(cfgs, Gen, Dis, ins, c["n_d"], c["seed"])
        exp_res = run_restatement(ocfg, y, GP, GB, DP, DB, ins, c["n_d"], c["seed"])
        worst = {}
        for k, v in exp_ref.items():
            fam = k.split("/")[0]
            a, b = v.double(), exp_res[k].double()
            err = float((a - b).abs().max() / (a.abs().max() + 1e-12))
            worst[fam] = max(worst.get(fam, 0.0), err)
            if compact:
                fix.update(compact_entries(k, v, c.get("sample", SAMPLE)))
            else:
                fix["exp/" + k] = v
        print(name, "restatement vs reference, max relative-to-range error per family:")
        for fam, e in sorted(worst.items()):
            print(f"   {fam:10s} {e:.3e}")
        assert max(worst.values()) < 2e-5, "oracle restatement disagrees with the reference"
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **{k: v.numpy() for k, v in fix.items()})
        with open(os.path.join(GOLDEN_DIR, name + ".json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("   wrote", name + ".npz", os.path.getsize(os.path.join(GOLDEN_DIR, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
