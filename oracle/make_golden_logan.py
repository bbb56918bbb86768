"""Golden vectors of LOGAN's latent optimisation written by the REAL reference (reference src/utils/losses.py:278-298 latent_optimise as src/utils/sample.py:123-135
calls it, the transport cost joining both losses at src/worker.py:319-321,598-599; configs/CIFAR10/LOGAN.yaml: unconditional ResNet generator with batch norm,
spectral-norm discriminator, hinge loss, uniform prior, lo_alpha 0.9, lo_beta 0.1, lo_lambda 0.1, lo_rate 0.8, lo_steps4train 2) at width 8: the initial networks,
the latents and the drop-mask draws, the discriminator update's loss / transport cost / gradients and the generator update's -- both back-propagate THROUGH the
gradient d D(G(z)) / dz (autograd's double backward over the reference's generator and discriminator). No restatement: the product is held against these vectors
(tests/aug_checks.py logan_case). Output: tests/golden/logan.npz (+ .json).

    python -m oracle.make_golden_logan           (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import importlib
import json
import os

import numpy as np
import torch

from . import make_golden as MG
from . import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "logan")
YAML = {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
        "MODEL": {"backbone": "resnet", "apply_d_sn": True, "z_prior": "uniform", "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8},
        "LOSS": {"adv_loss": "hinge", "apply_lo": True, "lo_alpha": 0.9, "lo_beta": 0.1, "lo_lambda": 0.1, "lo_rate": 0.8, "lo_steps4train": 2, "lo_steps4eval": 10},
        "OPTIMIZATION": {"batch_size": 4, "g_lr": 0.0002, "d_lr": 0.0002, "beta1": 0.5, "beta2": 0.999, "d_updates_per_step": 1}}
SEED, MASK_SEED = 2718, 31


def main():
    assert RI.available(), "needs the reference checkout"
    RI._prepare()
    misc = importlib.import_module("utils.misc")
    ref_losses = importlib.import_module("utils.losses")
    cfgs = RI.load_cfgs(YAML)
    cfgs.define_losses()
    torch.manual_seed(SEED)
    Gen, Dis = RI.build_models(cfgs)
    B, LS = YAML["OPTIMIZATION"]["batch_size"], cfgs.LOSS
    g = torch.Generator().manual_seed(SEED + 1)
    ins = {"real0": torch.randint(0, 256, (B, 3, 32, 32), generator=g).float() / 127.5 - 1.0, "rl0": torch.randint(0, 10, (B,), generator=g),
           "z0": torch.rand(B, 32, generator=g) * 2 - 1, "fl0": torch.randint(0, 10, (B,), generator=g),
           "z1": torch.rand(B, 32, generator=g) * 2 - 1, "fl1": torch.randint(0, 10, (B,), generator=g)}
    fix = {"in/" + k: v for k, v in ins.items()}
    for k, v in Gen.state_dict().items():
        fix["G_init/" + k] = v.clone()
    for k, v in Dis.state_dict().items():
        fix["D_init/" + k] = v.clone()

    def lo(z, fl, seed):
        torch.manual_seed(seed)
        fix[f"mask_draw/{seed}"] = torch.FloatTensor(B, 1).uniform_()
        torch.manual_seed(seed)
        return ref_losses.latent_optimise(zs=z, fake_labels=fl, generator=Gen, discriminator=Dis, batch_size=B, lo_rate=LS.lo_rate, lo_steps=LS.lo_steps4train,
                                          lo_alpha=LS.lo_alpha, lo_beta=LS.lo_beta, eval=False, cal_trsp_cost=True, device="cpu")
    # ---- discriminator update (src/worker.py:213-443 with src/utils/sample.py:123-135,162) ---------------------------------------------
    misc.make_GAN_trainable(Gen, None, Dis)
    misc.toggle_grad(Gen, False)
    misc.toggle_grad(Dis, True)
    Gen.apply(misc.untrack_bn_statistics)
    Dis.zero_grad()
    zs, cost = lo(ins["z0"], ins["fl0"], MASK_SEED)
    fake = Gen(zs, ins["fl0"])
    rd, fd = Dis(ins["real0"], ins["rl0"]), Dis(fake, ins["fl0"])
    d_loss = cfgs.LOSS.d_loss(rd["adv_output"], fd["adv_output"], DDP=False) + LS.lo_lambda * cost
    d_loss.backward()
    fix["d_loss"], fix["d_trsp_cost"], fix["d_zs"] = d_loss.detach().clone(), cost.detach().clone(), zs.detach().clone()
    for k, p in Dis.named_parameters():
        fix["D_grad/" + k] = p.grad.detach().clone()
    # ---- generator update (src/worker.py:502-634), on the networks as the discriminator update's forwards left them -----------------------
    misc.make_GAN_trainable(Gen, None, Dis)
    misc.toggle_grad(Dis, False)
    misc.toggle_grad(Gen, True)
    Gen.apply(misc.track_bn_statistics)
    Gen.zero_grad()
    zs, cost = lo(ins["z1"], ins["fl1"], MASK_SEED + 1)
    fake = Gen(zs, ins["fl1"])
    g_loss = cfgs.LOSS.g_loss(Dis(fake, ins["fl1"])["adv_output"], DDP=False) + LS.lo_lambda * cost
    g_loss.backward()
    fix["g_loss"], fix["g_trsp_cost"], fix["g_zs"] = g_loss.detach().clone(), cost.detach().clone(), zs.detach().clone()
    for k, p in Gen.named_parameters():
        fix["G_grad/" + k] = p.grad.detach().clone()
    np.savez_compressed(OUT + ".npz", **{k: v.detach().cpu().numpy() for k, v in fix.items()})
    json.dump({"yaml": YAML, "mask_seeds": [MASK_SEED, MASK_SEED + 1]}, open(OUT + ".json", "w"), indent=1)
    print(f"D loss {float(d_loss):.6e} (transport cost {float(cost):.4e})  G loss {float(g_loss):.6e}; wrote {OUT}.npz {os.path.getsize(OUT + '.npz') // 1024} KiB")


if __name__ == "__main__":
    main()
