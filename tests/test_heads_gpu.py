"""GPU: class-conditioning heads and losses (AC / TAC / ADC auxiliary classifiers, ContraGAN's 2C, ReACGAN's D2D-CE,
multi-hinge, multi-discriminator) of the mirrors against tests/golden/heads.npz -- the outputs of the REAL reference's
big_resnet.Discriminator + utils/losses.py combined as src/worker.py:281-317 does (oracle/make_golden_heads.py): head outputs, the
discriminator-side loss and every parameter gradient, fp32 <= 1e-3."""
import json
import os

import numpy as np
import pytest
import torch

from util import Collector
from test_model_gpu import build_from_yaml

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "heads.json")))


@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_heads_and_conditioning_losses_vs_reference(sg, name):
    import copy
    from studiogan_amd import losses as SL
    z = np.load(os.path.join(HERE, "golden", "heads.npz"))
    dev = torch.device("cuda:0")
    get = lambda k: torch.from_numpy(z[name + "/" + k])
    c, hp = META["cases"][name], META["hp"]
    y = c["yaml"]
    _, D = build_from_yaml(y, False, dev)
    sd = {k[len(name) + 3:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith(name + "/P/") or k.startswith(name + "/B/")}
    D.load_state_dict(sd, strict=True)
    D.train()
    real, fake, rl, fl = (get("in/" + k).to(dev) for k in ("real", "fake", "rl", "fl"))
    mtd, aux, nc = y["MODEL"]["d_cond_mtd"], y["MODEL"]["aux_cls_type"], y["DATA"]["num_classes"]
    ncl = nc * 2 if aux == "ADC" else nc
    cl = {"AC": lambda: SL.CrossEntropyLoss(),
          "2C": lambda: SL.ConditionalContrastiveLoss(num_classes=ncl, temperature=hp["temperature"], master_rank="cuda", DDP=False),
          "D2DCE": lambda: SL.Data2DataCrossEntropyLoss(num_classes=ncl, temperature=hp["temperature"], m_p=hp["m_p"], master_rank="cuda", DDP=False)
          }.get(mtd, lambda: None)()
    for p in D.parameters():
        p.grad = None
    rd = D(real, rl)
    fd = D(fake, fl, adc_fake=(aux == "ADC"))
    if c["adv_loss"] == "MH":
        lossy = torch.full((real.shape[0],), nc, dtype=torch.long, device=dev)
        loss = SL.crammer_singer_loss(DDP=False, **rd) + SL.crammer_singer_loss(fd["adv_output"], lossy, DDP=False)
    else:
        loss = SL.d_hinge(rd["adv_output"], fd["adv_output"], DDP=False)
    if cl is not None:
        loss = loss + hp["cond_lambda"] * cl(**rd)
        if aux == "TAC":
            loss = loss + hp["tac_dis_lambda"] * copy.deepcopy(cl)(**fd)
        elif aux == "ADC":
            loss = loss + hp["cond_lambda"] * cl(**fd)
    loss.backward()
    torch.cuda.synchronize()
    C = Collector()
    C.check("loss", loss.detach().reshape(1), get("exp/loss").reshape(1).float(), 1e-3)
    for k in ("adv_output", "cls_output", "embed", "proxy"):
        if (name + "/exp/real_" + k) in z.files:
            C.check("real " + k, rd[k], get("exp/real_" + k), 1e-3)
            C.check("fake " + k, fd[k], get("exp/fake_" + k), 1e-3)
    grads = {k[len(name) + 6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/grad/")}
    gm = max(float(v.abs().max()) for v in grads.values())
    for k, p in D.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        C.check("grad " + k, g, grads[k], 1e-3, floor=1e-2 * gm)
    C.finish()
