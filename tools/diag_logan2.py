"""GPU diagnostic, stage by stage: G(z), D(G(z)), d D(G(z)) / dz of the LOGAN fixture's networks against the fp64 CPU oracle (tests-only use of oracle/)."""
import importlib, json, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import studiogan_amd
from studiogan_amd import ops
from studiogan_amd.losses import cal_deriv
from studiogan_amd.worker import toggle_grad, untrack_bn_statistics, track_bn_statistics
from oracle import restate as O, make_golden as MG
from util import GOLDEN
dev = torch.device("cuda:0")
z = np.load(os.path.join(GOLDEN, "logan.npz")); meta = json.load(open(os.path.join(GOLDEN, "logan.json")))
y = meta["yaml"]; M, Dt, Ls = y["MODEL"], y["DATA"], y["LOSS"]
MODEL = types.SimpleNamespace(info_type="N/A", g_info_injection="N/A")
bb = importlib.import_module("studiogan_amd.backbones.resnet")
MOD = ops.Modules(apply_g_sn=False, apply_d_sn=True, g_cond_mtd="W/O", backbone="resnet")
G = bb.Generator(M["z_dim"], "N/A", Dt["img_size"], M["g_conv_dim"], False, ["N/A"], "W/O", Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
D = bb.Discriminator(Dt["img_size"], M["d_conv_dim"], True, False, ["N/A"], "W/O", "W/O", "N/A", False, Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
gsd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G_init/")}
dsd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("D_init/")}
gen, dis = O.model_fns(MG.oracle_cfg(y))
pnames_g = {k for k, _ in G.named_parameters()}; pnames_d = {k for k, _ in D.named_parameters()}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def run(mode, n_fwd, scales=None):
    G.load_state_dict({k: v.to(dev) for k, v in gsd.items()}, strict=True); D.load_state_dict({k: v.to(dev) for k, v in dsd.items()}, strict=True)
    G.train(); D.train(); toggle_grad(G, False); toggle_grad(D, True)
    G.apply(untrack_bn_statistics if mode == "untrack" else track_bn_statistics)
    GP = {k: v.double() for k, v in gsd.items() if k in pnames_g}; GB = {k: v.clone().double() if v.is_floating_point() else v.clone() for k, v in gsd.items() if k not in pnames_g}
    DP = {k: v.double() for k, v in dsd.items() if k in pnames_d}; DB = {k: v.clone().double() if v.is_floating_point() else v.clone() for k, v in dsd.items() if k not in pnames_d}
    for i in range(n_fwd):
        zi = torch.from_numpy(z["in/z0"]) * (scales[i] if scales else (1.0 - 0.1 * i)); fl = torch.from_numpy(z["in/fl0"])
        zc = zi.double().requires_grad_(True)
        img_o = gen(zc, fl, GP, GB, mode)
        adv_o, _ = dis(img_o, fl, DP, DB)
        zg_o = torch.autograd.grad(adv_o.sum(), zc, create_graph=True)[0]
        c_o = (zg_o.norm(2, dim=1) ** 2).mean()
        dg_o = torch.autograd.grad(c_o, [DP[k].requires_grad_(True) for k in DP], allow_unused=True) if False else None
        zd = zi.to(dev).requires_grad_(True)
        img = G(zd, fl.to(dev), eval=False)
        out = D(img, fl.to(dev), eval=False)
        zg = cal_deriv(inputs=zd, outputs=out["adv_output"], device=dev)
        torch.cuda.synchronize()
        print(f"{mode} fwd#{i}: image {rel(img, img_o):.2e}  logits {rel(out['adv_output'], adv_o):.2e}  z_grads {rel(zg, zg_o):.2e}", flush=True)


run("untrack", 3)
run("untrack", 4, [0.9, 1.0, 1.0, 1.001])
run("untrack", 3, [1.0001, 1.0, 0.9999])
