"""CPU: the augmentation oracle (oracle/aug_ref.py) against the vectors the reference's own apply_diffaug / apply_cr_aug / MSELoss wrote
(tests/golden/aug.npz), the regeneration of those vectors from the reference when it is present, the host mirrors' argument behaviour, and the kernel
SOURCES of csrc/ext/augment.hip run lane by lane on the CPU interpreter (tests/hipemu) through the product's own Python layer against the same vectors --
the checks tests/test_aug_gpu.py makes on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
import emu  # noqa: E402
import aug_checks as AC  # noqa: E402
from oracle import aug_ref as AR  # noqa: E402
from oracle import make_golden_aug as MGA  # noqa: E402
from oracle import ref_import  # noqa: E402

_NT = torch.get_num_threads()      # (the interpreter fixture below pins torch to one thread for the module: reference-side runs put the default back)
needs_emu = pytest.mark.skipif(not emu.available(), reason="host clang++ of the ROCm toolchain not found")


def test_aug_oracle_reproduces_the_reference_vectors():
    z = np.load(AC.GOLD)
    for tag, shape, policy in MGA.DIFFAUG_CASES:
        p = f"diffaug/{tag}/"
        draws = []
        while p + f"draw{len(draws)}" in z.files:
            draws.append(AC._t(z, p + f"draw{len(draws)}"))
        x = AC._t(z, p + "x").requires_grad_(True)
        y = AR.diffaug(x, policy, draws)
        assert torch.equal(y.detach(), AC._t(z, p + "y")), tag
        gy = AC._t(z, p + "gy").requires_grad_(True)
        (dx,) = torch.autograd.grad(y, x, gy, create_graph=True)
        assert float((dx.detach() - AC._t(z, p + "dx")).abs().max()) <= 1e-6, tag
        (lin,) = torch.autograd.grad(dx, gy, AC._t(z, p + "gg"))
        assert float((lin - AC._t(z, p + "lin")).abs().max()) <= 1e-6, tag
    for tag, shape, flip, trans in MGA.CR_CASES:
        p = f"cr/{tag}/"
        x = AC._t(z, p + "x").requires_grad_(True)
        y = AR.cr_aug(x, AC._t(z, p + "coin"), AC._t(z, p + "tx"), AC._t(z, p + "ty"))
        assert torch.equal(y.detach(), AC._t(z, p + "y")), tag
        assert float((torch.autograd.grad(y, x, AC._t(z, p + "gy"))[0] - AC._t(z, p + "dx")).abs().max()) <= 1e-6, tag
    for tag, shape in MGA.MSE_CASES:
        p = f"mse/{tag}/"
        assert abs(float(AR.mse(AC._t(z, p + "a"), AC._t(z, p + "b"))) - float(z[p + "loss"])) <= 1e-7


def test_aug_draw_order_matches_the_reference_consumption():
    """draw_diffaug / draw_cr (what the host mirrors call) leave the generator in the state the reference's functions leave it in"""
    if not ref_import.available():
        pytest.skip("the reference checkout is only present in the authoring container")
    import importlib
    ref_import._prepare()
    RD, RC = importlib.import_module("utils.diffaug"), importlib.import_module("utils.cr")
    x = torch.randn(3, 3, 16, 16)
    for policy in ("color,translation,cutout", "cutout,translation", "color"):
        torch.manual_seed(5)
        RD.apply_diffaug(x, policy)
        a = torch.rand(4)
        torch.manual_seed(5)
        AR.draw_diffaug(x.shape, policy)
        assert torch.equal(a, torch.rand(4)), policy
    torch.manual_seed(6)
    RC.apply_cr_aug(x)
    a = torch.rand(4)
    torch.manual_seed(6)
    AR.draw_cr(x.shape)
    assert torch.equal(a, torch.rand(4))


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_aug_golden_vectors_regenerate_from_the_reference(tmp_path, monkeypatch):
    out = tmp_path / "aug.npz"
    monkeypatch.setattr(MGA, "OUT", str(out))
    MGA.main()
    a, b = np.load(AC.GOLD), np.load(out)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_aug_host_mirrors_refuse_the_cpu_and_bad_arguments():
    import studiogan_amd  # noqa: F401
    from studiogan_amd import diffaug, cr, losses
    x = torch.randn(2, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        diffaug.apply_diffaug(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cr.apply_cr_aug(x)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        losses.l2_loss(x, x)
    assert diffaug.apply_diffaug(x, policy="") is x and cr.apply_cr_aug(x, flip=False, translation=False) is x       # diffaug.py:36, cr.py:18


def test_ada_apa_heuristic_matches_the_reference_formula():
    """worker.adapt_aa_p against the reference's lines (src/worker.py:478-482) evaluated literally with its tensor arithmetic"""
    import studiogan_amd  # noqa: F401
    from studiogan_amd.worker import adapt_aa_p
    rng = np.random.default_rng(0)
    for _ in range(200):
        aa_p, target, kimg = float(rng.uniform(0, 1)), float(rng.choice([0.6, 0.5, 0.9])), float(rng.choice([100, 500]))
        cnt = float(rng.choice([64, 256, 1024]) * rng.integers(1, 9))
        ssum = float(rng.integers(-int(cnt), int(cnt) + 1))
        dis_sign_real = torch.tensor((ssum, cnt))
        heuristic = (dis_sign_real[0] / dis_sign_real[1]).item()
        adjust = np.sign(heuristic - target) * (dis_sign_real[1].item()) / (kimg * 1000)
        ref = min(torch.as_tensor(1.), max(aa_p + adjust, torch.as_tensor(0.)))
        assert abs(adapt_aa_p(aa_p, ssum, cnt, target, kimg) - float(ref)) <= 1e-7


# ---- the kernel sources on the interpreter ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def installed():
    import fullemu
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    with fullemu.Installed(dma_late=1, greedy=1, seed=3) as E:
        yield E
    torch.set_num_threads(n)


@needs_emu
@pytest.mark.parametrize("case", AC.DIFFAUG_CASES, ids=[c[0] for c in AC.DIFFAUG_CASES])
def test_emulated_diffaug_matches_reference_vectors(installed, case):
    c0 = installed.counters()["launches"]
    AC.diffaug_case(case, torch.device("cpu"))
    assert installed.counters()["launches"] > c0


@needs_emu
@pytest.mark.parametrize("case", AC.CR_CASES, ids=[c[0] for c in AC.CR_CASES])
def test_emulated_cr_aug_matches_reference_vectors(installed, case):
    AC.cr_case(case, torch.device("cpu"))


@needs_emu
@pytest.mark.parametrize("case", AC.MSE_CASES, ids=[c[0] for c in AC.MSE_CASES])
def test_emulated_l2_loss_matches_reference_vectors(installed, case):
    AC.mse_case(case, torch.device("cpu"))


@needs_emu
def test_emulated_further_losses_match_reference_vectors(installed):
    for kind in AC.LOSS_KINDS:
        AC.loss_case(kind, torch.device("cpu"))
    for case in AC.FM_CASES:
        AC.fm_case(case, torch.device("cpu"))


@needs_emu
def test_emulated_apa_select_and_weight_clipping(installed):
    for i in range(3):
        AC.apa_case(i, torch.device("cpu"))
    AC.clamp_case(torch.device("cpu"))


@needs_emu
def test_emulated_augment_operator_subsets_and_properties(installed):
    from studiogan_amd import _lib as L
    dev = torch.device("cpu")
    every = [L.AUG_BRIGHTNESS, L.AUG_SATURATION, L.AUG_CONTRAST, L.AUG_FLIP, L.AUG_CUTOUT]
    k = 0
    for tr in (0, L.AUG_TRANSLATE, L.AUG_TRANSLATE_REFLECT):
        for sub in range(1 << len(every)):
            ops = tr | sum(b for i, b in enumerate(every) if sub >> i & 1)
            if ops and (sub % 3 == k % 3):            # a third of the 96 subsets per translation kind, rotating: every operator pair meets
                AC.oracle_spec_case((2, 3, 8, 12) if k % 2 else (3, 3, 9, 7), ops, dev, seed=k)
            k += 1
    allz = L.AUG_BRIGHTNESS | L.AUG_SATURATION | L.AUG_CONTRAST | L.AUG_FLIP | L.AUG_CUTOUT
    AC.adjoint_and_linearity((3, 3, 16, 16), allz | L.AUG_TRANSLATE, dev, 1)
    AC.adjoint_and_linearity((2, 3, 10, 14), allz | L.AUG_TRANSLATE_REFLECT, dev, 2)
    AC.adjoint_and_linearity((2, 1, 8, 8), L.AUG_CONTRAST | L.AUG_TRANSLATE_REFLECT, dev, 3)


@needs_emu
@pytest.mark.parametrize("tag", AC.CONSISTENCY_CASES)
def test_emulated_worker_update_with_diffaug_and_consistency_regularisers(installed, tag):
    """the worker's discriminator and generator update with DiffAugment + bCR + zCR (BigGAN), CR and DiffAugment alone (SNGAN) through the interpreted
    kernel sources: loss and every gradient against the REAL reference's (tests/golden/consistency.npz)"""
    AC.consistency_case(tag, torch.device("cpu"))


@needs_emu
def test_emulated_r1_through_diffaug(installed):
    AC.r1_through_diffaug_case("biggan32", torch.device("cpu"))


@needs_emu
def test_emulated_ada_pipeline_matches_reference_vectors(installed):
    from oracle import make_golden_ada as MGD
    for case in MGD.CASES:
        AC.ada_case(case, torch.device("cpu"))
    AC.ada_adjoint_case((2, 3, 12, 10), torch.device("cpu"), 1)
    AC.ada_adjoint_case((3, 1, 9, 16), torch.device("cpu"), 2)
    AC.ada_filter_adjoint_case((2, 3, 24, 26), torch.device("cpu"), 3)
    AC.ada_filter_adjoint_case((2, 1, 9, 12), torch.device("cpu"), 4)


@needs_emu
@pytest.mark.parametrize("tag", AC.INFO_CASES)
def test_emulated_infogan_updates_match_reference_vectors(installed, tag):
    """InfoGAN: generator code injection ("cBN" / "concat"), the discriminator's Q heads, the information losses and the Q heads' Adam with the generator's
    settings, through the interpreted kernel sources against the REAL reference (tests/golden/info.npz)"""
    if tag in ("sngan32_info_cbn", "bigdeep32_info_cbn", "dcgan32_info_cbn") and os.environ.get("SG_EMU_NET") != "1":
        pytest.skip("SG_EMU_NET=1 runs the remaining InfoGAN fixtures through the interpreter (full-width DCGAN: ~40 s)")
    AC.info_case(tag, torch.device("cpu"))


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_info_golden_vectors_regenerate_from_the_reference(tmp_path, monkeypatch):
    from oracle import make_golden_info as MGI
    from util import GOLDEN
    monkeypatch.setattr(MGI, "OUT", str(tmp_path / "info"))
    nt = torch.get_num_threads()
    torch.set_num_threads(_NT)
    MGI.main()
    a, b = np.load(os.path.join(GOLDEN, "info.npz")), np.load(tmp_path / "info.npz")
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    torch.set_num_threads(nt)


@needs_emu
def test_emulated_r1_on_the_mdgan_discriminator(installed):
    """configs/*/MDGAN.yaml: R1 on a BigGAN discriminator with attention and the multi-discriminator head (the adversarial logit through LinearFn -> LinearDgradFn)"""
    AC.r1_with_heads_case("md", torch.device("cpu"))


@needs_emu
def test_emulated_freeze_d(installed):
    AC.freeze_d_case("sngan32", torch.device("cpu"), 2)


@needs_emu
def test_emulated_logan_latent_optimisation(installed):
    """LOGAN: both updates back-propagate through d D(G(z)) / dz (the create_graph pass through the GENERATOR: LinearDgradFn, ConvDgradFn with the fused upsampling,
    batch-norm double backward, TanhGradFn) against the REAL reference's double backward (tests/golden/logan.npz)"""
    AC.logan_case(torch.device("cpu"))


def test_emulated_logan_discriminator_side_against_the_oracle(installed):
    """the GPU suite's tight discriminator-side LOGAN check (fp64 oracle double backward, scaled latents) on the interpreter"""
    AC.logan_oracle_case(torch.device("cpu"), 0.9)


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_logan_golden_vectors_regenerate_from_the_reference(tmp_path, monkeypatch):
    from oracle import make_golden_logan as MGL
    from util import GOLDEN
    monkeypatch.setattr(MGL, "OUT", str(tmp_path / "logan"))
    nt = torch.get_num_threads()
    torch.set_num_threads(_NT)
    MGL.main()
    a, b = np.load(os.path.join(GOLDEN, "logan.npz")), np.load(tmp_path / "logan.npz")
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    torch.set_num_threads(nt)


@pytest.mark.parametrize("name", AC.STANDING_CASES)
def test_emulated_generator_preparation_for_evaluation(installed, name):
    """worker.GeneratorController.prepare_generator: standing statistics (five training-mode forwards over batches of 1..7 latents), batch statistics, plain
    evaluation -- buffers, training flags and the evaluation image against the REAL reference's (tests/golden/standing.npz)"""
    if name != "biggan" and os.environ.get("SG_EMU_NET") != "1":
        pytest.skip("SG_EMU_NET=1 runs the other two generators through the interpreter as well")
    AC.standing_case(name, torch.device("cpu"))


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_standing_golden_vectors_regenerate_from_the_reference(tmp_path, monkeypatch):
    from oracle import make_golden_standing as MGS
    from util import GOLDEN
    monkeypatch.setattr(MGS, "OUT", str(tmp_path / "standing"))
    nt = torch.get_num_threads()
    torch.set_num_threads(_NT)
    MGS.main()
    a, b = np.load(os.path.join(GOLDEN, "standing.npz")), np.load(tmp_path / "standing.npz")
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    torch.set_num_threads(nt)


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_emulated_logan_latent_optimisation_at_evaluation_time(installed, monkeypatch):
    """metrics.generate_images_and_stack_features(latent_opt=...): LOGAN's step of the latents at EVALUATION time (reference src/utils/sample.py:96,123-135 with
    LOSS.lo_steps4eval; generator and discriminator in eval mode, the create_graph pass through batch norm on running statistics) against the REAL reference's
    sample.generate_images(is_train=False) under the same seed: the latents the generator finally receives and its images. (lo_alpha raised so that the step is visible:
    at this width the latent gradient is ~6e7 and the configured step ~1e-9.)"""
    import copy
    import importlib
    import types
    from oracle import make_golden_logan as MGL
    from studiogan_amd import config_map as CM, metrics as M
    ref_import._prepare()
    sample = importlib.import_module("utils.sample")
    y = MGL.YAML
    nt = torch.get_num_threads()
    torch.set_num_threads(_NT)
    cfgs = ref_import.load_cfgs(y)
    cfgs.define_losses()
    cfgs.LOSS.lo_alpha, cfgs.LOSS.lo_beta = 2e7, 0.1
    torch.manual_seed(3)
    Gr, Dr = ref_import.build_models(cfgs)
    with torch.no_grad():
        for _ in range(2):          # running statistics worth evaluating with
            Gr(torch.rand(4, 32) * 2 - 1, torch.randint(0, 10, (4,)))
    gs, ds = copy.deepcopy(Gr.state_dict()), copy.deepcopy(Dr.state_dict())
    Gr.eval(), Dr.eval()
    got = {}

    class Cap(torch.nn.Module):
        def __init__(self, g):
            super().__init__()
            self.g = g

        def forward(self, zs, ys, eval=False):
            got["zs"] = zs.detach().clone()
            return self.g(zs, ys, eval=eval)
    torch.manual_seed(21)
    img_r = sample.generate_images(z_prior="uniform", truncation_factor=-1.0, batch_size=4, z_dim=32, num_classes=10, y_sampler="totally_random", radius="N/A", generator=Cap(Gr),
                                   discriminator=Dr, is_train=False, LOSS=cfgs.LOSS, RUN=types.SimpleNamespace(langevin_sampling=False), MODEL=cfgs.MODEL, device="cpu",
                                   is_stylegan=False, generator_mapping=None, generator_synthesis=None, style_mixing_p=0.0, stylegan_update_emas=False, cal_trsp_cost=False)[0].detach()
    zs_r = got["zs"]
    torch.manual_seed(21)
    torch.randint(low=0, high=10, size=(4,), dtype=torch.long)
    z0 = torch.FloatTensor(4, 32).uniform_(-1.0, 1.0)
    torch.set_num_threads(nt)
    assert float((zs_r - z0).abs().max()) > 1e-2          # the reference's latents did move

    class Stub:
        def get_outputs(self, x, quantize=True):
            got["img"] = x.detach().clone()
            return torch.zeros(x.shape[0], 8), torch.zeros(x.shape[0], 8)
    monkeypatch.setattr(M, "softmax_rows", lambda t: t)
    dev = torch.device("cpu")
    G, D, _ = CM.build(y, dev)
    G.load_state_dict(gs, strict=True), D.load_state_dict(ds, strict=True)
    G.eval(), D.eval()
    torch.manual_seed(21)
    LS = y["LOSS"]
    M.generate_images_and_stack_features(Cap(G), Stub(), 4, 4, 32, 10, device=dev, z_prior="uniform",
                                         latent_opt=dict(discriminator=D, lo_rate=LS["lo_rate"], lo_steps=LS["lo_steps4eval"], lo_alpha=2e7, lo_beta=0.1))
    step_err = float(((got["zs"] - z0) - (zs_r - z0)).abs().max() / (zs_r - z0).abs().max())
    assert step_err <= 1e-4, step_err
    assert float((got["img"] - img_r).abs().max()) <= 1e-5


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_emulated_langevin_sampling_matches_the_reference(installed):
    """losses.langevin_sampling (reference src/utils/sample.py:195-216, RUN.langevin_sampling): four steps with a decay after every second one on a conditional
    generator + projection discriminator in eval mode, against the REAL reference's function under the same seed (the noise comes from the same torch.distributions
    objects). The rate is tiny because the energy gradient of these width-8 networks is ~1e9."""
    import copy
    import importlib
    from studiogan_amd import config_map as CM, losses as SL
    ref_import._prepare()
    sample = importlib.import_module("utils.sample")
    y = {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
         "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8}}
    nt = torch.get_num_threads()
    torch.set_num_threads(_NT)
    cfgs = ref_import.load_cfgs(y)
    torch.manual_seed(3)
    Gr, Dr = ref_import.build_models(cfgs)
    with torch.no_grad():
        for _ in range(40):
            Gr(torch.randn(8, 32), torch.randint(0, 10, (8,)))
    gs, ds = copy.deepcopy(Gr.state_dict()), copy.deepcopy(Dr.state_dict())
    Gr.eval(), Dr.eval()
    g = torch.Generator().manual_seed(5)
    z0, lab = torch.randn(4, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
    kw = dict(z_dim=32, fake_labels=lab, batch_size=4, langevin_rate=1e-11, langevin_noise_std=0.1, langevin_decay=0.5, langevin_decay_steps=2, langevin_steps=4)
    torch.manual_seed(9)
    zr = sample.langevin_sampling(zs=z0.clone(), generator=Gr, discriminator=Dr, device="cpu", **kw).detach()
    torch.set_num_threads(nt)
    dev = torch.device("cpu")
    G, D, _ = CM.build(y, dev)
    G.load_state_dict(gs, strict=True), D.load_state_dict(ds, strict=True)
    G.eval(), D.eval()
    torch.manual_seed(9)
    zm = SL.langevin_sampling(zs=z0.clone(), generator=G, discriminator=D, device=dev, **kw).detach()
    move = float((zr - z0).abs().max())
    assert move > 1e-2
    assert float((zm - zr).abs().max()) <= 1e-4 * move, (float((zm - zr).abs().max()), move)


@pytest.mark.parametrize("name", ["ReACGAN-ADC-DiffAug", "BigGAN-Info", "WGAN-GP"])
def test_emulated_config_steps_against_the_references_worker(installed, name):
    """tests/golden/config_steps.npz (all 55 CIFAR10 files on the GPU: tests/test_wide_zz_config_steps_gpu.py): the draws the reference's own worker consumed, replayed
    into this package's Worker on networks rebuilt from the seed -- three files here (all 55 ran on the interpreter when the fixture was written)"""
    if name not in AC.config_step_names():
        pytest.skip("fixture without this file")
    AC.config_step_case(name, torch.device("cpu"))


def test_consistency_oracle_reproduces_the_reference_vectors():
    """oracle/restate.py d_consistency_loss / g_consistency_loss on the committed networks and draws == the reference's values in the fixture"""
    import json
    from util import load_golden, sub, GOLDEN
    from oracle import restate as O, make_golden as MG
    z = np.load(os.path.join(GOLDEN, "consistency.npz"))
    meta_c = json.load(open(os.path.join(GOLDEN, "consistency.json")))
    for tag, m in meta_c.items():
        hp = m["hp"]
        if hp.get("ada_type"):
            continue          # (no restatement of the ADA pipeline: the product is held against the reference's vectors directly)
        fix, meta = load_golden(m["config"])
        y = meta["yaml"]
        ocfg = MG.oracle_cfg(y)
        gen_fn, dis_fn = O.model_fns(ocfg)

        def split(pre, names):
            st = sub(fix, pre)
            P = {k: st[k].clone() for k in names}
            return P, {k: v.clone() for k, v in st.items() if k not in P}
        GP, GB = split("G_init/", [k[len(tag) + 8:] for k in z.files if k.startswith(tag + "/G_grad/")])
        DP, DB = split("D_init/", [k[len(tag) + 8:] for k in z.files if k.startswith(tag + "/D_grad/")])
        p = tag + "/"

        def lst(prefix, name):
            return [torch.from_numpy(z[f"{p}{prefix}/{name}/{i}"]) if f"{p}{prefix}/{name}/{i}" in z.files else None for i in range(8)]
        draws = {}
        if hp.get("diffaug_policy"):
            draws["series_real"] = [t for t in lst("draw_d", "series_real") if t is not None]
            draws["series_fake"] = [t for t in lst("draw_d", "series_fake") if t is not None]
        for name in ("prl_real", "prl_fake"):
            if f"{p}draw_d/{name}/0" in z.files:
                draws[name] = lst("draw_d", name)[:3]
        if f"{p}draw_d/apa/0" in z.files:
            draws["apa"] = lst("draw_d", "apa")[:1]
        ins = sub(fix, "in/")
        zed = torch.from_numpy(z[p + "z_eps_d"]) if p + "z_eps_d" in z.files else None
        leaves = O._leaves(DP)
        loss, _ = O.d_consistency_loss(gen_fn, dis_fn, GP, GB, leaves, DB, ocfg, ins["real0"], ins["rl0"], ins["z0"], ins["fl0"], hp.get("adv_loss", y["LOSS"]["adv_loss"]), hp, draws, zed)
        loss.backward()
        assert abs(float(loss.detach()) - float(z[p + "d_loss"])) <= 1e-5 * abs(float(z[p + "d_loss"])), tag
        gmax = max(float(np.abs(z[p + "D_grad/" + k]).max()) for k in leaves)
        for k, v in leaves.items():
            assert float((v.grad - torch.from_numpy(z[p + "D_grad/" + k])).abs().max()) <= 1e-4 * gmax, (tag, k)      # (torch's own CPU sums move by 2.5e-5 with the thread count)


@pytest.mark.skipif(not ref_import.available(), reason="the reference checkout is only present in the authoring container")
def test_consistency_golden_vectors_regenerate_from_the_reference(tmp_path, monkeypatch):
    from oracle import make_golden_consistency as MGC
    monkeypatch.setattr(MGC, "OUT", str(tmp_path / "consistency"))
    nt = torch.get_num_threads()
    torch.set_num_threads(_NT)
    import shutil
    from util import GOLDEN
    for name in ("biggan32", "sngan32"):
        shutil.copy(os.path.join(GOLDEN, name + ".npz"), tmp_path / (name + ".npz"))
    MGC.main()
    a, b = np.load(os.path.join(GOLDEN, "consistency.npz")), np.load(tmp_path / "consistency.npz")
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    torch.set_num_threads(nt)
