"""Golden vectors of the training step's discriminator-side and generator-side losses WITH DiffAugment and the consistency regularisers
(CR / bCR / zCR), written by the REAL reference: its models (models/big_resnet.py, models/resnet.py), utils/diffaug.apply_diffaug as
cfgs.AUG.series_augment, utils/cr.apply_cr_aug as cfgs.AUG.parallel_augment, torch.nn.MSELoss as the worker's l2_loss, combined exactly as
src/worker.py:236-365 (discriminator) and :520-603 (generator) combine them, under a seeded generator. Also the pin of the restatement
(oracle/restate.py d_consistency_loss / g_consistency_loss with oracle/aug_ref.py fed the same draws) against them.
Networks and inputs are those of tests/golden/<config>.npz (same seeds: asserted), so this fixture only adds the draws and the expected values.
Output: tests/golden/consistency.npz (+ .json).

    python -m oracle.make_golden_consistency        (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import importlib
import json
import os

import numpy as np
import torch

from . import aug_ref as AR
from . import make_golden as MG
from . import ref_import as RI
from . import restate as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "consistency")

CASES = {     # tag -> (configuration of oracle/make_golden.py, hyper-parameters; the lambdas / radius of reference src/configs/CIFAR10/BigGAN-{CR,ICR,DiffAug}.yaml)
    "biggan32_diffaug_bcr_zcr": ("biggan32", dict(diffaug_policy="color,translation,cutout", bcr_lambdas=(10.0, 10.0), d_lambda=20.0, g_lambda=0.5, radius=0.05)),
    "sngan32_cr": ("sngan32", dict(cr_lambda=10.0)),
    "sngan32_diffaug": ("sngan32", dict(diffaug_policy="translation,cutout")),
    # least-squares adversarial loss (configs/CIFAR10/LSGAN.yaml) + feature matching in the generator update (LOSS.apply_fm of configs/*/MHGAN.yaml) behind DiffAugment
    "sngan32_ls_fm_diffaug": ("sngan32", dict(adv_loss="least_square", fm_lambda=1.0, diffaug_policy="translation,cutout")),
    # adaptive pseudo augmentation at p = 0.5 (configs/CIFAR10/SNGAN-APA.yaml) + weight clipping after the discriminator's optimiser step (configs/CIFAR10/WGAN-WC.yaml: 0.01)
    "sngan32_apa_wc": ("sngan32", dict(apa_p=0.5, wc_bound=0.01, adv_loss="wasserstein")),
    # adaptive discriminator augmentation, 'bgc' pipeline at p = 0.8 (configs/CIFAR10/SNGAN-ADA.yaml). No restatement of the pipeline exists: the fixture carries
    # the draws the reference's AdaAugment made (recorded at torch.rand / torch.randn) and the product is held against the reference directly
    "sngan32_ada": ("sngan32", dict(ada_type="bgc", ada_p=0.8)),
}
AUG_SEED = 31337


def draw_all(hp, shape):
    """the generator's draws in the order the reference worker's discriminator step consumes them (series(real), series(fake), then the parallel views)"""
    d = {}
    if hp.get("apa_p") is not None:
        d["apa"] = [torch.rand([shape[0], 1, 1, 1])]
    if hp.get("diffaug_policy"):
        d["series_real"] = AR.draw_diffaug(shape, hp["diffaug_policy"])
        d["series_fake"] = AR.draw_diffaug(shape, hp["diffaug_policy"])
    if hp.get("cr_lambda") is not None:
        d["prl_real"] = AR.draw_cr(shape)
    if hp.get("bcr_lambdas") is not None:
        d["prl_real"] = AR.draw_cr(shape)
        d["prl_fake"] = AR.draw_cr(shape)
    return d


def reference_d_side(cfgs, Gen, Dis, ins, hp, z_eps, RD, RC, misc):
    l2 = torch.nn.MSELoss()                                                              # src/worker.py:116
    series = (lambda x: RD.apply_diffaug(x, hp["diffaug_policy"])) if hp.get("diffaug_policy") else (lambda x: x)      # config.py:586-587 / misc.identity
    if hp.get("_ada") is not None:
        series = hp["_ada"]                                                               # config.py:590-591
    misc.make_GAN_trainable(Gen, None, Dis)
    misc.toggle_grad(Gen, False)
    misc.toggle_grad(Dis, True)
    Gen.apply(misc.untrack_bn_statistics)
    Dis.zero_grad()
    real, rl, fl = ins["real0"], ins["rl0"], ins["fl0"]
    fake = Gen(ins["z0"], fl)                                                             # utils/sample.py:162
    fake_eps = Gen(z_eps, fl) if z_eps is not None else None                              # utils/sample.py:176
    torch.manual_seed(AUG_SEED)
    if hp.get("apa_p") is not None:                                                       # worker.py:273-274
        real = importlib.import_module("utils.apa_aug").apply_apa_aug(real, fake.detach(), hp["apa_p"], "cpu")
    real_, fake_ = series(real), series(fake)                                             # worker.py:277-278
    rd, fd = Dis(real_, rl), Dis(fake_, fl)
    loss = cfgs.LOSS.d_loss(rd["adv_output"], fd["adv_output"], DDP=False)
    if hp.get("cr_lambda") is not None:                                                   # worker.py:325-336
        rp = Dis(RC.apply_cr_aug(real), rl)
        loss = loss + hp["cr_lambda"] * l2(rd["adv_output"], rp["adv_output"])
    if hp.get("bcr_lambdas") is not None:                                                 # worker.py:339-354
        real_prl, fake_prl = RC.apply_cr_aug(real), RC.apply_cr_aug(fake)
        rp, fp = Dis(real_prl, rl), Dis(fake_prl, fl)
        loss = loss + hp["bcr_lambdas"][0] * l2(rd["adv_output"], rp["adv_output"]) + hp["bcr_lambdas"][1] * l2(fd["adv_output"], fp["adv_output"])
    if hp.get("d_lambda") is not None:                                                    # worker.py:357-365
        fe = Dis(fake_eps, fl)
        loss = loss + hp["d_lambda"] * l2(fd["adv_output"], fe["adv_output"])
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in Dis.named_parameters()}
    after = None
    if hp.get("wc_bound") is not None:                                                    # worker.py:440-443,489-492: optimiser step, then the clip
        before = {k: p.detach().clone() for k, p in Dis.named_parameters()}
        cfgs.define_optimizer(Gen, Dis)
        cfgs.OPTIMIZATION.d_optimizer.step()
        for p in Dis.parameters():
            p.data.clamp_(-hp["wc_bound"], hp["wc_bound"])
        after = {k: p.detach().clone() for k, p in Dis.named_parameters()}
        with torch.no_grad():                                                             # (the generator side of the fixture is taken on the un-stepped weights)
            for k, p in Dis.named_parameters():
                p.copy_(before[k])
    return loss.detach().clone(), grads, fake.detach().clone(), after


def reference_g_side(cfgs, Gen, Dis, ins, hp, z_eps, RD, misc):
    ref_losses = importlib.import_module("utils.losses")
    l2 = torch.nn.MSELoss()
    misc.make_GAN_trainable(Gen, None, Dis)
    misc.toggle_grad(Dis, False)
    misc.toggle_grad(Gen, True)
    Gen.apply(misc.track_bn_statistics)
    Gen.zero_grad()
    fl = ins["fl1"]
    fake = Gen(ins["z1"], fl)
    fake_eps = Gen(z_eps, fl) if z_eps is not None else None
    torch.manual_seed(AUG_SEED + 1)
    fake_ = RD.apply_diffaug(fake, hp["diffaug_policy"]) if hp.get("diffaug_policy") else fake          # worker.py:549-550
    if hp.get("_ada") is not None:
        fake_ = hp["_ada"](fake)
    fake_dict = Dis(fake_, fl)
    loss = cfgs.LOSS.g_loss(fake_dict["adv_output"], DDP=False)
    if hp.get("fm_lambda") is not None:                                                   # worker.py:588-596
        real_ = RD.apply_diffaug(ins["real1"], hp["diffaug_policy"]) if hp.get("diffaug_policy") else ins["real1"]
        real_dict = Dis(real_, ins["rl1"])
        loss = loss + hp["fm_lambda"] * ref_losses.feature_matching_loss(real_dict["h"].detach(), fake_dict["h"])
    if hp.get("g_lambda") is not None:                                                    # worker.py:601-603
        loss = loss + hp["g_lambda"] * (-1 * l2(fake, fake_eps))
    loss.backward()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in Gen.named_parameters()}, fake.detach().clone()


def main():
    assert RI.available(), "needs the reference checkout"
    RI._prepare()
    RD, RC = importlib.import_module("utils.diffaug"), importlib.import_module("utils.cr")
    misc = importlib.import_module("utils.misc")
    fix, meta = {}, {}
    for tag, (cfg_name, hp) in CASES.items():
        c = MG.CONFIGS[cfg_name]
        y = c["yaml"]
        if hp.get("adv_loss"):
            y = dict(y, LOSS=dict(y.get("LOSS", {}), adv_loss=hp["adv_loss"]))
        cfgs = RI.load_cfgs(y)
        cfgs.define_losses()
        torch.manual_seed(c["seed"])
        Gen, Dis = RI.build_models(cfgs)
        gold = np.load(os.path.join(os.path.dirname(OUT), cfg_name + ".npz"))
        for k, v in Gen.state_dict().items():
            assert np.array_equal(gold["G_init/" + k], v.numpy()), k                      # the committed fixture's networks: nothing to store twice
        GP, GB = RI.split_state(Gen)
        DP, DB = RI.split_state(Dis)
        ocfg = MG.oracle_cfg(y)
        ins = MG.synth_inputs(c["seed"] + 1, c["n_d"], c["batch"], ocfg["z_dim"], ocfg["num_classes"], ocfg["img_size"])
        shape = tuple(ins["real0"].shape)
        g = torch.Generator().manual_seed(c["seed"] + 7)
        ze_d = ins["z0"] + hp["radius"] * torch.randn(ins["z0"].shape, generator=g) if hp.get("d_lambda") is not None else None      # utils/sample.py:80-82
        ze_g = ins["z1"] + hp["radius"] * torch.randn(ins["z1"].shape, generator=g) if hp.get("g_lambda") is not None else None
        p = tag + "/"
        # ---- discriminator side -------------------------------------------------------------------------------------------------
        torch.manual_seed(AUG_SEED)
        draws = draw_all(hp, shape)
        ada = hp.get("ada_type")
        if ada:
            from . import make_golden_ada as MGD
            RAda = importlib.import_module("utils.ada_aug")
            mod = RAda.AdaAugment(**MGD.PIPES[ada]).train().requires_grad_(False)
            mod.p.copy_(torch.as_tensor(hp["ada_p"]))
            hp = dict(hp, _ada=mod)
            with MGD.Recorded() as rec:
                d_loss, d_grads, fake0, d_after = reference_d_side(cfgs, Gen, Dis, ins, hp, ze_d, RD, RC, misc)
            for i, t in enumerate(rec.draws):
                fix[p + f"draw_d/ada/{i:03d}"] = t
            fix[p + "d_loss"] = d_loss
            for k, v in d_grads.items():
                fix[p + "D_grad/" + k] = v
            with MGD.Recorded() as rec:
                g_loss, g_grads, fake1 = reference_g_side(cfgs, Gen, Dis, ins, hp, ze_g, RD, misc)
            for i, t in enumerate(rec.draws):
                fix[p + f"draw_g/ada/{i:03d}"] = t
            fix[p + "g_loss"] = g_loss
            for k, v in g_grads.items():
                fix[p + "G_grad/" + k] = v
            print(f"{tag:28s} reference only: D loss {float(d_loss):.8e}, G loss {float(g_loss):.8e}")
            meta[tag] = {"config": cfg_name, "hp": {k: v for k, v in hp.items() if not k.startswith("_")}}
            continue
        d_loss, d_grads, fake0, d_after = reference_d_side(cfgs, Gen, Dis, ins, hp, ze_d, RD, RC, misc)
        if d_after is not None:
            for k, v in d_after.items():
                fix[tag + "/D_after/" + k] = v
        gen_fn, dis_fn = O.model_fns(ocfg)
        leaves = O._leaves(DP)
        o_loss, o_fake = O.d_consistency_loss(gen_fn, dis_fn, GP, GB, leaves, DB, ocfg, ins["real0"], ins["rl0"], ins["z0"], ins["fl0"],
                                              y["LOSS"]["adv_loss"], hp, draws, ze_d)
        o_loss.backward()
        gmax = max(float(v.abs().max()) for v in d_grads.values())
        e_v = abs(float(o_loss.detach()) - float(d_loss)) / abs(float(d_loss))
        e_g = max(float((leaves[k].grad - v).abs().max()) for k, v in d_grads.items()) / gmax
        e_f = float((o_fake - fake0).abs().max())
        print(f"{tag:28s} D side: reference {float(d_loss):.8e} restatement {float(o_loss.detach()):.8e}  rel.err value {e_v:.2e} grads {e_g:.2e} fake {e_f:.2e}")
        assert e_v < 2e-6 and e_g < 2e-5 and e_f < 1e-5, tag
        fix[p + "d_loss"] = d_loss
        for k, v in d_grads.items():
            fix[p + "D_grad/" + k] = v
        for name, lst in draws.items():
            for i, t in enumerate(lst):
                if t is not None:
                    fix[p + f"draw_d/{name}/{i}"] = t
        if ze_d is not None:
            fix[p + "z_eps_d"] = ze_d
        # ---- generator side (the networks have advanced by the discriminator side's forwards on both implementations) -------------
        torch.manual_seed(AUG_SEED + 1)
        draws_g = {"series_fake": AR.draw_diffaug(shape, hp["diffaug_policy"])} if hp.get("diffaug_policy") else {}
        if hp.get("fm_lambda") is not None:
            if hp.get("diffaug_policy"):
                draws_g["series_real_fm"] = AR.draw_diffaug(shape, hp["diffaug_policy"])
            hp = dict(hp, _fm_real=(ins["real1"], ins["rl1"]))
        g_loss, g_grads, fake1 = reference_g_side(cfgs, Gen, Dis, ins, hp, ze_g, RD, misc)
        gl = O._leaves(GP)
        og_loss, og_fake = O.g_consistency_loss(gen_fn, dis_fn, gl, GB, DP, DB, ocfg, ins["z1"], ins["fl1"], y["LOSS"]["adv_loss"], hp, draws_g, ze_g)
        og_loss.backward()
        gmax = max(float(v.abs().max()) for v in g_grads.values())
        e_v = abs(float(og_loss.detach()) - float(g_loss)) / max(abs(float(g_loss)), 1e-30)
        e_g = max(float((gl[k].grad - v).abs().max()) for k, v in g_grads.items()) / gmax
        print(f"{'':28s} G side: reference {float(g_loss):.8e} restatement {float(og_loss.detach()):.8e}  rel.err value {e_v:.2e} grads {e_g:.2e}")
        assert e_v < 2e-5 and e_g < 5e-5, tag
        fix[p + "g_loss"] = g_loss
        for k, v in g_grads.items():
            fix[p + "G_grad/" + k] = v
        for name in ("series_fake", "series_real_fm"):
            for i, t in enumerate(draws_g.get(name, [])):
                fix[p + f"draw_g/{name}/{i}"] = t
        if ze_g is not None:
            fix[p + "z_eps_g"] = ze_g
        meta[tag] = {"config": cfg_name, "hp": {k: v for k, v in hp.items() if not k.startswith("_")}}
    np.savez_compressed(OUT + ".npz", **{k: v.detach().cpu().numpy() for k, v in fix.items()})
    json.dump(meta, open(OUT + ".json", "w"), indent=1)
    print("wrote", OUT + ".npz", os.path.getsize(OUT + ".npz") // 1024, "KiB")


if __name__ == "__main__":
    main()
