"""GPU diagnostic for the LOGAN discriminator-side mismatch (r05 driver run): moved latents of losses.latent_optimise against the fixture with the generator
frozen / trainable, twice each (determinism)."""
import importlib, json, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import studiogan_amd
from studiogan_amd import ops, losses as SL
from studiogan_amd.worker import toggle_grad, make_GAN_trainable, untrack_bn_statistics
from aug_checks import Replayed
from util import GOLDEN
dev = torch.device("cuda:0")
z = np.load(os.path.join(GOLDEN, "logan.npz")); meta = json.load(open(os.path.join(GOLDEN, "logan.json")))
y = meta["yaml"]; M, Dt, Ls = y["MODEL"], y["DATA"], y["LOSS"]
MODEL = types.SimpleNamespace(info_type="N/A", g_info_injection="N/A")
bb = importlib.import_module("studiogan_amd.backbones.resnet")
MOD = ops.Modules(apply_g_sn=False, apply_d_sn=True, g_cond_mtd="W/O", backbone="resnet")
G = bb.Generator(M["z_dim"], "N/A", Dt["img_size"], M["g_conv_dim"], False, ["N/A"], "W/O", Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
D = bb.Discriminator(Dt["img_size"], M["d_conv_dim"], True, False, ["N/A"], "W/O", "W/O", "N/A", False, Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
gsd = {k[7:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("G_init/")}
dsd = {k[7:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("D_init/")}
s0 = meta["mask_seeds"][0]
zs0, fl0 = torch.from_numpy(z["in/z0"]).to(dev), torch.from_numpy(z["in/fl0"]).to(dev)
want = torch.from_numpy(z["d_zs"]).double()


def run(g_train, tag):
    G.load_state_dict(gsd, strict=True); D.load_state_dict(dsd, strict=True)
    G.train(); D.train()
    toggle_grad(G, g_train); toggle_grad(D, True)
    G.apply(untrack_bn_statistics)
    with Replayed([torch.from_numpy(z[f"mask_draw/{s0}"])]):
        zs, cost = SL.latent_optimise(zs=zs0, fake_labels=fl0, generator=G, discriminator=D, batch_size=4, lo_rate=Ls["lo_rate"], lo_steps=Ls["lo_steps4train"],
                                      lo_alpha=Ls["lo_alpha"], lo_beta=Ls["lo_beta"], eval=False, cal_trsp_cost=True, device=dev)
    torch.cuda.synchronize()
    e = (zs.detach().double().cpu() - want).abs().max() / want.abs().max()
    c = abs(float(cost) - float(z["d_trsp_cost"])) / float(z["d_trsp_cost"])
    print(f"{tag}: moved latents err {float(e):.3e}  transport cost err {c:.3e}", flush=True)
    return zs.detach().clone()


a = run(False, "G frozen   #1")
b = run(False, "G frozen   #2")
c = run(True, "G trainable #1")
d = run(True, "G trainable #2")
print("frozen #1 vs #2 max diff", float((a - b).abs().max()), " trainable #1 vs #2", float((c - d).abs().max()), " frozen vs trainable", float((a - c).abs().max()))
