"""Pin the class-conditioning heads / losses of the oracle (oracle/restate.py d_heads, cond_loss, crammer_singer, d_side_loss) against the
REAL reference: models/big_resnet.py Discriminator with d_cond_mtd in {AC, 2C, D2DCE, MH, MD} and aux_cls_type in {W/O, TAC, ADC},
utils/losses.py CrossEntropyLoss / ConditionalContrastiveLoss / Data2DataCrossEntropyLoss / crammer_singer_loss, combined the way
src/worker.py:281-317 combines them on the discriminator side. Writes tests/golden/heads.npz (+ .json): per configuration the initial
state, inputs, and the reference's loss, head outputs and every parameter gradient.

    python oracle/make_golden_heads.py        (authoring container only)
TEST INFRASTRUCTURE ONLY."""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import as R      # noqa: E402
from oracle import restate as O         # noqa: E402
from oracle import make_golden as MG    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "heads")
HP = {"cond_lambda": 0.7, "temperature": 0.5, "m_p": 0.98, "tac_dis_lambda": 0.6}
BASE = {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
        "MODEL": {"backbone": "big_resnet", "g_cond_mtd": "cBN", "apply_g_sn": True, "apply_d_sn": True, "apply_attn": True, "attn_g_loc": [2],
                  "attn_d_loc": [1], "z_dim": 40, "g_shared_dim": 16, "g_conv_dim": 8, "d_conv_dim": 8},
        "LOSS": {"adv_loss": "hinge"}, "OPTIMIZATION": {"batch_size": 8}}
CASES = {
    "ac": dict(d_cond_mtd="AC", aux_cls_type="W/O"),
    "ac_norm": dict(d_cond_mtd="AC", aux_cls_type="W/O", normalize_d_embed=True),
    "ac_tac": dict(d_cond_mtd="AC", aux_cls_type="TAC"),
    "ac_adc": dict(d_cond_mtd="AC", aux_cls_type="ADC"),
    "2c": dict(d_cond_mtd="2C", aux_cls_type="W/O", d_embed_dim=16, normalize_d_embed=True),
    "d2dce": dict(d_cond_mtd="D2DCE", aux_cls_type="W/O", d_embed_dim=16, normalize_d_embed=True),
    "d2dce_adc": dict(d_cond_mtd="D2DCE", aux_cls_type="ADC", d_embed_dim=16, normalize_d_embed=True),
    "d2dce_tac_raw": dict(d_cond_mtd="D2DCE", aux_cls_type="TAC", d_embed_dim=16, normalize_d_embed=False),
    "mh": dict(d_cond_mtd="MH", aux_cls_type="W/O", adv_loss="MH"),
    "md": dict(d_cond_mtd="MD", aux_cls_type="W/O"),
}


def main():
    assert R.available()
    import copy
    fix, meta = {}, {"hp": HP, "cases": {}}
    ref_losses = None
    for name, c in CASES.items():
        y = copy.deepcopy(BASE)
        y["MODEL"].update({k: v for k, v in c.items() if k != "adv_loss"})
        adv = c.get("adv_loss", "hinge")
        y["LOSS"]["adv_loss"] = adv
        cfgs = R.load_cfgs(y)
        ref_losses = importlib.import_module("utils.losses")
        torch.manual_seed(100 + len(fix))
        _, Dis = R.build_models(cfgs)
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():      # non-initial state: attention gate on, random biases
            for k, p in Dis.named_parameters():
                if k.endswith("sigma"):
                    p.fill_(0.6)
                elif k.endswith(".bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        P, B = R.split_state(Dis)
        nc, bs = 10, 8
        real = torch.randint(0, 256, (bs, 3, 32, 32), generator=g).float() / 127.5 - 1.0
        fake = torch.tanh(torch.randn(bs, 3, 32, 32, generator=g))
        rl = torch.randint(0, nc, (bs,), generator=g)
        fl = torch.randint(0, nc, (bs,), generator=g)
        rl[1] = rl[0]; fl[2] = fl[0]            # guaranteed same-class pairs (positive terms of the contrastive losses)
        # ---- the reference, combined as src/worker.py:281-317 does
        mtd, aux = c["d_cond_mtd"], c["aux_cls_type"]
        ncl = nc * 2 if aux == "ADC" else nc
        if mtd == "AC":
            cl = ref_losses.CrossEntropyLoss()
        elif mtd == "2C":
            cl = ref_losses.ConditionalContrastiveLoss(num_classes=ncl, temperature=HP["temperature"], master_rank="cpu", DDP=False)
        elif mtd == "D2DCE":
            cl = ref_losses.Data2DataCrossEntropyLoss(num_classes=ncl, temperature=HP["temperature"], m_p=HP["m_p"], master_rank="cpu", DDP=False)
        else:
            cl = None
        Dis.train()
        rd = Dis(real, rl)
        fd = Dis(fake, fl, adc_fake=(aux == "ADC"))
        if adv == "MH":
            lossy = torch.full((bs,), nc, dtype=torch.long)
            loss = ref_losses.crammer_singer_loss(DDP=False, **rd) + ref_losses.crammer_singer_loss(fd["adv_output"], lossy, DDP=False)
        else:
            loss = ref_losses.d_hinge(rd["adv_output"], fd["adv_output"], DDP=False)
        if cl is not None:
            loss = loss + HP["cond_lambda"] * cl(**rd)
            if aux == "TAC":
                loss = loss + HP["tac_dis_lambda"] * copy.deepcopy(cl)(**fd)
            elif aux == "ADC":
                loss = loss + HP["cond_lambda"] * cl(**fd)
        loss.backward()
        ref_grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in Dis.named_parameters()}
        # ---- the restatement on the same state
        ocfg = dict(MG.oracle_cfg(y), num_classes=nc)
        leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        Bc = {k: v.clone() for k, v in B.items()}
        lo, rdo, fdo = O.d_side_loss(O.model_fns(ocfg)[1], leaves, Bc, ocfg, real, rl, fake, fl, adv, HP)
        lo.backward()
        worst = abs(float(lo) - float(loss))
        for k in ref_grads:
            go = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
            worst = max(worst, float((go - ref_grads[k]).abs().max()))
        for k in ("adv_output", "cls_output", "embed", "proxy"):
            if rd[k] is not None:
                worst = max(worst, float((rd[k].detach() - rdo[k].detach()).abs().max()))
        print(f"{name:14s} loss {float(loss):.6f}  max |restatement - reference| = {worst:.2e}")
        assert worst < 5e-6, name
        pre = name + "/"
        for k, v in P.items():
            fix[pre + "P/" + k] = v.numpy()
        for k, v in B.items():
            fix[pre + "B/" + k] = v.numpy()
        for k, v in (("real", real), ("fake", fake), ("rl", rl), ("fl", fl)):
            fix[pre + "in/" + k] = v.numpy()
        fix[pre + "exp/loss"] = np.float64(float(loss))
        for k in ("adv_output", "cls_output", "embed", "proxy"):
            if rd[k] is not None:
                fix[pre + "exp/real_" + k] = rd[k].detach().numpy()
                fix[pre + "exp/fake_" + k] = fd[k].detach().numpy()
        for k, v in ref_grads.items():
            fix[pre + "grad/" + k] = v.numpy()
        meta["cases"][name] = {"yaml": y, "adv_loss": adv}
    np.savez_compressed(OUT + ".npz", **fix)
    json.dump(meta, open(OUT + ".json", "w"), indent=1)
    print("wrote", OUT + ".npz", os.path.getsize(OUT + ".npz") // 1024, "KiB")


if __name__ == "__main__":
    main()
