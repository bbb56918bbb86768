"""Pin the regulariser restatements of oracle/restate.py (R1, max-gradient penalty, DRAGAN penalty, LeCam, top-k, the uint8 input
transform) against the REAL reference functions (reference src/utils/losses.py:262-366, src/utils/ops.py:106-133, the
ToTensor+Normalize transform of src/data_util.py:92-94) run here on CPU, and write tests/golden/regularisers.npz with the inputs and
the reference's outputs. tests/test_oracle_cpu.py re-checks the restatement against this fixture on every CPU run.

    python oracle/make_golden_regularisers.py         (authoring container only: imports /root/reference through oracle/ref_import.py)

TEST INFRASTRUCTURE ONLY.
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import as RI      # noqa: E402
from oracle import restate as O          # noqa: E402
from oracle import make_golden as MG     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "regularisers")
CFG = "sngp32"      # SN discriminator with projection head (the double-backward path through SN, PD head, pooling, ReLU)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    spec = MG.CONFIGS[CFG]
    y = spec["yaml"]
    cfgs = RI.load_cfgs(y)
    ref_losses = importlib.import_module("utils.losses")
    ref_ops = importlib.import_module("utils.ops")
    torch.manual_seed(spec["seed"])
    _, Dis = RI.build_models(cfgs)
    Dis.train()
    DP, DB = RI.split_state(Dis)
    ocfg = MG.oracle_cfg(y)
    dis_fn = O.model_fns(ocfg)[1]
    g = torch.Generator().manual_seed(4711)
    B, S, ncls = 4, y["DATA"]["img_size"], y["DATA"]["num_classes"]
    real = torch.randint(0, 256, (B, 3, S, S), generator=g).float() / 127.5 - 1.0
    fake = torch.randint(0, 256, (B, 3, S, S), generator=g).float() / 127.5 - 1.0
    lab = torch.randint(0, ncls, (B,), generator=g)
    fix = {"in/real": real, "in/fake": fake, "in/lab": lab}
    for k, v in DP.items():
        fix["D_P/" + k] = v
    for k, v in DB.items():
        fix["D_B/" + k] = v
    report = {}

    def grads_of(loss, params):
        params = list(params)
        gs = torch.autograd.grad(loss, params, allow_unused=True)
        return [torch.zeros_like(p) if g_ is None else g_ for p, g_ in zip(params, gs)]

    def run(name, ref_fn, ora_fn):
        # reference: on the real module (its SN buffers advance: restore them afterwards); restatement: on copies of the same state
        state = {k: v.clone() for k, v in Dis.state_dict().items()}
        r = ref_fn()
        rg = grads_of(r, Dis.parameters())
        Dis.load_state_dict(state)
        leaves = {k: v.clone().requires_grad_(True) for k, v in DP.items()}
        o = ora_fn(leaves, {k: v.clone() for k, v in DB.items()})
        og = grads_of(o, leaves.values())
        err_v = float((o - r).detach().abs() / max(float(r.detach().abs()), 1e-30))
        gmax = max(float(a.abs().max()) for a in rg)
        err_g = max(float((a - b).abs().max()) for a, b in zip(rg, og)) / max(gmax, 1e-30)
        report[name] = {"value_rel_err": err_v, "grad_rel_err": err_g}
        print(f"{name:10s} reference {float(r):.8e}  restatement {float(o):.8e}  rel.err value {err_v:.2e}  grads {err_g:.2e}")
        assert err_v < 1e-6 and err_g < 1e-5, name
        fix["exp/" + name] = r.detach()
        for (k, _), a in zip(Dis.named_parameters(), rg):
            fix[f"exp/{name}_grad/" + k] = a.detach()

    # ---- max-gradient penalty (losses.py:338-352): alpha = torch.rand(B, 1) on the host RNG ------------------------------------
    def ref_maxgp():
        torch.manual_seed(11)
        return ref_losses.cal_maxgrad_penalty(real_images=real, real_labels=lab, fake_images=fake, discriminator=Dis, device="cpu")
    torch.manual_seed(11)
    alpha = torch.rand(B, 1)
    fix["in/maxgp_alpha"] = alpha
    run("maxgp", ref_maxgp, lambda P, Bf: O.maxgrad_penalty(dis_fn, real, lab, fake, P, Bf, alpha))

    # ---- DRAGAN penalty (:319-335): alpha = torch.rand(B,1,1,1), then torch.rand(real.size()) ----------------------------------
    def ref_dra():
        torch.manual_seed(12)
        return ref_losses.cal_dra_penalty(real_images=real, real_labels=lab, discriminator=Dis, device="cpu")
    torch.manual_seed(12)
    a4, noise = torch.rand(B, 1, 1, 1), torch.rand(real.size())
    fix["in/dra_alpha"], fix["in/dra_noise"] = a4, noise
    run("dra", ref_dra, lambda P, Bf: O.dra_penalty(dis_fn, real, lab, P, Bf, a4, noise))

    # ---- R1 (:355-361) through the same forward that feeds the loss (worker.py:260-261,410-412) ----------------------------------
    def ref_r1():
        x = real.clone().requires_grad_(True)
        out = Dis(x, lab)
        return ref_losses.cal_r1_reg(adv_output=out["adv_output"], images=x, device="cpu")
    run("r1", ref_r1, lambda P, Bf: O.r1_reg(dis_fn, real, lab, P, Bf)[0])

    # ---- LeCam (:262-265) with the reference's EMA object (ops.py:106-133) --------------------------------------------------------
    lr_, lf_ = torch.randn(37, generator=g), torch.randn(37, generator=g)
    ema = ref_ops.LeCamEMA()
    ema.decay, ema.start_itr = 0.9, 2
    seq = [(0.5, "D_real", 0), (-0.25, "D_fake", 0), (1.0, "D_real", 5), (0.0, "D_fake", 5)]
    for cur, mode, itr in seq:
        ema.update(cur, mode, itr)
    a, b = lr_.clone().requires_grad_(True), lf_.clone().requires_grad_(True)
    lc = ref_losses.lecam_reg(a, b, ema)
    lc.backward()
    a2, b2 = lr_.clone().requires_grad_(True), lf_.clone().requires_grad_(True)
    lo = O.lecam_reg(a2, b2, ema.D_real, ema.D_fake)
    lo.backward()
    assert torch.equal(lc.detach(), lo.detach()) and torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad), "lecam"
    fix.update({"in/lecam_real": lr_, "in/lecam_fake": lf_, "exp/lecam": lc.detach(), "exp/lecam_dreal": a.grad, "exp/lecam_dfake": b.grad,
                "exp/lecam_ema": torch.tensor([ema.D_real, ema.D_fake], dtype=torch.float64)})
    print("lecam      bit-identical; EMA after the update sequence:", ema.D_real, ema.D_fake)

    # ---- top-k (worker.py:565-566) and adjust_k (:364-366) ---------------------------------------------------------------------
    x = torch.randn(64, generator=g)
    fix["in/topk_x"], fix["exp/topk_10"] = x, torch.topk(x, 10).values
    ks, k = [], 64
    for _ in range(100):
        k = ref_losses.adjust_k(current_k=k, topk_gamma=0.99, inf_k=int(64 * 0.5))
        ks.append(k)
    fix["exp/adjust_k"] = torch.tensor(ks, dtype=torch.float64)

    # ---- uint8 input transform (data_util.py:92-94: ToTensor + Normalize([0.5]*3, [0.5]*3)); torchvision is absent here, so the
    # two transforms are applied by their definition: ToTensor = HWC uint8 -> CHW float32 .div(255), Normalize = (x - mean) / std
    xu = torch.randint(0, 256, (4, 16, 16, 3), generator=g, dtype=torch.uint8)
    t = xu.permute(0, 3, 1, 2).contiguous().to(dtype=torch.float32).div(255)
    mean = torch.as_tensor([0.5, 0.5, 0.5]).view(-1, 1, 1)
    std = torch.as_tensor([0.5, 0.5, 0.5]).view(-1, 1, 1)
    t = t.sub(mean).div(std)
    assert torch.equal(O.uint8_to_normalized(xu), t), "uint8 transform"
    fix["in/u8"], fix["exp/u8_norm"] = xu, t

    np.savez_compressed(OUT + ".npz", **{k: v.detach().cpu().numpy() for k, v in fix.items()})
    json.dump({"config": CFG, "yaml": y, "report": report,
               "lecam_updates": seq, "note": "reference functions of src/utils/losses.py / src/utils/ops.py run on CPU by oracle/make_golden_regularisers.py"},
              open(OUT + ".json", "w"), indent=1)
    print("wrote", OUT + ".npz", os.path.getsize(OUT + ".npz") // 1024, "KiB")


if __name__ == "__main__":
    main()
