"""Golden vectors of InfoGAN training written by the REAL reference: its generator with the code injection (models/big_resnet.py:81-92,125-130 "cBN";
models/resnet.py:95-107,142-147 "concat"), its discriminator with the Q heads (models/big_resnet.py:337-344,373-377), the code sampling of utils/sample.py:113-118,
the parameter toggling and losses of src/worker.py:220-224,508-512,607-618 (torch.nn.CrossEntropyLoss per discrete code, utils/losses.normal_nll_loss) and the
optimiser split of src/config.py:499-517 (the Q heads' parameters in the GENERATOR's Adam). Per case: the initial networks, the inputs and codes, the
discriminator update's loss and gradients, the generator update's loss and gradients (generator AND Q heads), and the Q-head parameters after the generator's Adam
step. No restatement: the product is held against these vectors directly (tests/aug_checks.py info_case). Output: tests/golden/info.npz (+ .json).

    python -m oracle.make_golden_info           (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import importlib
import json
import os

import numpy as np
import torch

from . import make_golden as MG
from . import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "info")

CASES = {
    # BigGAN with both kinds of codes through the conditional batch norms (configs/CIFAR10/BigGAN-Info.yaml uses "discrete" + "cBN"; "both" exercises every head)
    "biggan32_info_cbn": dict(base="biggan32", MODEL=dict(info_type="both", g_info_injection="cBN", info_num_discrete_c=2, info_dim_discrete_c=5, info_num_conti_c=3),
                              LOSS=dict(infoGAN_loss_discrete_lambda=1.0, infoGAN_loss_conti_lambda=0.5)),
    # SNGAN (ResNet backbone) with one discrete code through a mixing layer in front of z
    "sngan32_info_concat": dict(base="sngan32", MODEL=dict(info_type="discrete", g_info_injection="concat", info_num_discrete_c=1, info_dim_discrete_c=10),
                                LOSS=dict(infoGAN_loss_discrete_lambda=1.0)),
    # the configuration files' own setting: one 10-way code through the conditional batch norms of the ResNet generator (configs/CIFAR10/SNGAN-Info.yaml)
    "sngan32_info_cbn": dict(base="sngan32", MODEL=dict(info_type="discrete", g_info_injection="cBN", info_num_discrete_c=1, info_dim_discrete_c=10),
                             LOSS=dict(infoGAN_loss_discrete_lambda=1.0)),
    # BigGAN-deep (the reference's legacy variant carries the same code paths, models/big_resnet_deep_legacy.py:110-121,156-161,322-329)
    "bigdeep32_info_cbn": dict(base="bigdeep32", MODEL=dict(info_type="discrete", g_info_injection="cBN", info_num_discrete_c=1, info_dim_discrete_c=10),
                               LOSS=dict(infoGAN_loss_discrete_lambda=1.0)),
    # DCGAN with the configuration file's setting (configs/CIFAR10/DCGAN-Info.yaml): full-width deep_conv networks, batch-norm discriminator
    # (compact: initial state by formula, oracle/make_golden.py formula_state; large expected tensors as 4096 evenly spaced samples + their l2 norm)
    "dcgan32_info_cbn": dict(base="dcgan32", compact=True, MODEL=dict(info_type="discrete", g_info_injection="cBN", info_num_discrete_c=1, info_dim_discrete_c=10),
                             LOSS=dict(infoGAN_loss_discrete_lambda=1.0)),
}
INFO_PARAMS = ["info_discrete_linear", "info_conti_mu_linear", "info_conti_var_linear"]      # reference src/config.py:346


def main():
    assert RI.available(), "needs the reference checkout"
    RI._prepare()
    misc = importlib.import_module("utils.misc")
    ref_losses = importlib.import_module("utils.losses")
    fix, meta = {}, {}
    for tag, c in CASES.items():
        base = MG.CONFIGS[c["base"]]
        y = {k: dict(v) for k, v in base["yaml"].items()}
        y["MODEL"].update(c["MODEL"])
        y.setdefault("LOSS", {}).update(c["LOSS"])
        cfgs = RI.load_cfgs(y)
        cfgs.define_losses()
        torch.manual_seed(base["seed"] + 17)
        Gen, Dis = RI.build_models(cfgs)
        compact = bool(c.get("compact"))
        if compact:
            Gen.load_state_dict(MG.formula_state({k: list(v.shape) for k, v in Gen.state_dict().items()}, base["seed"] + 31), strict=True)
            Dis.load_state_dict(MG.formula_state({k: list(v.shape) for k, v in Dis.state_dict().items()}, base["seed"] + 32), strict=True)
        M = cfgs.MODEL
        ocfg = MG.oracle_cfg(y)
        ins = MG.synth_inputs(base["seed"] + 1, 1, base["batch"], ocfg["z_dim"], ocfg["num_classes"], ocfg["img_size"])
        B = base["batch"]
        g = torch.Generator().manual_seed(base["seed"] + 23)
        p = tag + "/"
        if not compact:
            for k, v in Gen.state_dict().items():
                fix[p + "G_init/" + k] = v.clone()
            for k, v in Dis.state_dict().items():
                fix[p + "D_init/" + k] = v.clone()
        codes = {}
        for side in ("d", "g"):       # utils/sample.py:113-118
            disc = torch.randint(M.info_dim_discrete_c, (B, M.info_num_discrete_c), generator=g) if M.info_type in ("discrete", "both") else None
            conti = torch.rand(B, M.info_num_conti_c, generator=g) * 2 - 1 if M.info_type in ("continuous", "both") else None
            codes[side] = (disc, conti)
            if disc is not None:
                fix[p + f"code_{side}_disc"] = disc
            if conti is not None:
                fix[p + f"code_{side}_conti"] = conti

        def with_codes(z, side):
            disc, conti = codes[side]
            if disc is not None:
                z = torch.cat((z, torch.nn.functional.one_hot(disc, M.info_dim_discrete_c).view(B, -1)), dim=1)
            if conti is not None:
                z = torch.cat((z, conti), dim=1)
            return z
        info_mods = [getattr(Dis, n) for n in INFO_PARAMS if hasattr(Dis, n)]
        cfgs.define_optimizer(Gen, Dis)                                                   # src/config.py:499-565: Q heads in the generator's Adam
        # ---- discriminator update (src/worker.py:213-443) ----------------------------------------------------------------------------
        misc.make_GAN_trainable(Gen, None, Dis)
        misc.toggle_grad(Gen, False)
        misc.toggle_grad(Dis, True)
        for m in info_mods:                                                               # worker.py:220-224
            misc.toggle_grad(m, grad=False, num_freeze_layers=-1, is_stylegan=False)
        Gen.apply(misc.untrack_bn_statistics)
        cfgs.OPTIMIZATION.d_optimizer.zero_grad()
        fake = Gen(with_codes(ins["z0"], "d"), ins["fl0"])
        rd, fd = Dis(ins["real0"], ins["rl0"]), Dis(fake, ins["fl0"])
        d_loss = cfgs.LOSS.d_loss(rd["adv_output"], fd["adv_output"], DDP=False)
        d_loss.backward()
        fix[p + "d_loss"] = d_loss.detach().clone()
        for k, prm in Dis.named_parameters():
            fix[p + "D_grad/" + k] = torch.zeros_like(prm) if prm.grad is None else prm.grad.detach().clone()
        before = {k: v.detach().clone() for k, v in Dis.named_parameters()}
        cfgs.OPTIMIZATION.d_optimizer.step()
        for k, prm in Dis.named_parameters():
            if any(n in k for n in INFO_PARAMS):
                assert torch.equal(prm.detach(), before[k]), k                            # not the discriminator optimiser's to move
            fix[p + "D_after_d/" + k] = prm.detach().clone()
        # ---- generator update (src/worker.py:502-634) -----------------------------------------------------------------------------------
        misc.make_GAN_trainable(Gen, None, Dis)
        misc.toggle_grad(Dis, False)
        misc.toggle_grad(Gen, True)
        for m in info_mods:                                                               # worker.py:508-512
            misc.toggle_grad(m, grad=True, num_freeze_layers=-1, is_stylegan=False)
        Gen.apply(misc.track_bn_statistics)
        cfgs.OPTIMIZATION.g_optimizer.zero_grad()
        for prm in Dis.parameters():
            prm.grad = None
        fake = Gen(with_codes(ins["z1"], "g"), ins["fl1"])
        fd = Dis(fake, ins["fl1"])
        g_loss = cfgs.LOSS.g_loss(fd["adv_output"], DDP=False)
        disc, conti = codes["g"]
        ce = torch.nn.CrossEntropyLoss()                                                  # worker.py:115
        if M.info_type in ("discrete", "both"):                                           # worker.py:607-615
            dim, tot = M.info_dim_discrete_c, 0.0
            for info_c in range(M.info_num_discrete_c):
                tot = tot + ce(fd["info_discrete_c_logits"][:, info_c * dim: dim * (info_c + 1)], disc[:, info_c: info_c + 1].squeeze())
            g_loss = g_loss + cfgs.LOSS.infoGAN_loss_discrete_lambda * tot
        if M.info_type in ("continuous", "both"):                                         # worker.py:616-618
            g_loss = g_loss + cfgs.LOSS.infoGAN_loss_conti_lambda * ref_losses.normal_nll_loss(conti, fd["info_conti_mu"], fd["info_conti_var"])
        g_loss.backward()
        fix[p + "g_loss"] = g_loss.detach().clone()
        for k, prm in Gen.named_parameters():
            fix[p + "G_grad/" + k] = prm.grad.detach().clone()
        for k, prm in Dis.named_parameters():
            if any(n in k for n in INFO_PARAMS):
                fix[p + "Q_grad/" + k] = prm.grad.detach().clone()
        cfgs.OPTIMIZATION.g_optimizer.step()
        for k, prm in Dis.named_parameters():
            if any(n in k for n in INFO_PARAMS):
                fix[p + "Q_after_g/" + k] = prm.detach().clone()
        for k, v in ins.items():
            fix[p + "in/" + k] = v
        if compact:          # big tensors: samples + norm
            for k in [k for k in fix if k.startswith(p) and fix[k].numel() > 16384 and not k.startswith(p + "in/")]:
                v = fix.pop(k).reshape(-1).float()
                fix[k + "#s"] = v[MG.sample_index(v.numel(), 4096)].clone()
                fix[k + "#n"] = torch.tensor([float(v.double().norm())], dtype=torch.float64)
        meta[tag] = {"yaml": y, "batch": B, "compact": compact, "seeds": [base["seed"] + 31, base["seed"] + 32]}
        print(f"{tag:22s} D loss {float(d_loss):.6e}  G loss (adversarial + information) {float(g_loss):.6e}  Q-head tensors: {sum(1 for k in fix if k.startswith(p + 'Q_grad/'))}")
    np.savez_compressed(OUT + ".npz", **{k: v.detach().cpu().numpy() for k, v in fix.items()})
    json.dump(meta, open(OUT + ".json", "w"), indent=1)
    print("wrote", OUT + ".npz", os.path.getsize(OUT + ".npz") // 1024, "KiB")


if __name__ == "__main__":
    main()
