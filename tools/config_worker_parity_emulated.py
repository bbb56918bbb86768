"""Training-step parity of EVERY non-StyleGAN configuration file against the reference's OWN worker code: the unmodified `WORKER.train_discriminator` and
`WORKER.train_generator` of reference src/worker.py:213-681 are run on the CPU (a WORKER object built by the reference's constructor with local_rank="cpu", the
reference's Configurations / define_losses / define_augments / define_optimizer, the reference's networks, Adam and EMA) and, from the same initial state, the same real
batches and the same torch seed, this package's `worker.Worker` built through `config_map.build` runs its two methods with the kernel SOURCES on the CPU interpreter.
Both consume the CPU generator in the same order (latents, labels, InfoGAN codes, DiffAugment / ADA / CR / APA draws, gradient-penalty interpolation, LOGAN's masks), so
no draw is injected or replayed; the only adaptation is that this package's latent sampler is switched to the reference's order of its two draws (labels first).

Compared per configuration: the discriminator loss of the last of TWO discriminator updates (the first one's Adam step lies in between), every discriminator parameter
gradient of that update and every discriminator parameter after both steps (+ weight clipping); the generator loss, every generator parameter gradient and every
generator parameter after its step; the ADA / APA probability after the heuristic. Channel widths cut to 8; image sizes, class counts, heads, losses, regularisers and
augmentations as the file says (ADA / APA strength raised from the files' 0.0 so that the pipelines actually fire).
   usage: python tools/config_worker_parity_emulated.py [--dir=CIFAR10] [--batch=4] [--nd=2] [--acml=1] [--steps=1] [--freezeD=-1] [--seed=77] [--emit=config_steps] [--verbose] [name ...]        TEST INFRASTRUCTURE; needs /root/reference."""
import copy
import glob
import importlib
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

N_D = 2


def rel(a, b, floor=0.0):
    a, b = a.detach().double().reshape(-1), b.detach().double().reshape(-1)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))


def worst(mine, ref, frac=1e-2):
    top = max(float(v.abs().max()) for v in ref.values())
    w, where = 0.0, ""
    for k, g in ref.items():
        e = rel(mine[k], g, floor=frac * top)
        if e > w:
            w, where = e, k
    return w, where


def grads(net):
    return {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()}


def params(net):
    return {k: p.detach().clone() for k, p in net.named_parameters()}


class RecordedDraws:
    """every torch.rand / randn / randint / FloatTensor(...).uniform_() result in call order (--emit: the draws the reference's step consumed, for tests/golden/config_steps.npz)"""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self.saved = (torch.rand, torch.randn, torch.randint, torch.FloatTensor)
        draws = self.draws

        def wrap(fn):
            def inner(*a, **k):
                t = fn(*a, **k)
                draws.append(t.detach().clone().cpu())
                return t
            return inner
        FT0 = torch.FloatTensor

        class FT:
            def __init__(self, *size):
                self.t = FT0(*size)

            def uniform_(self, a=0.0, b=1.0):
                t = self.t.uniform_(a, b)
                draws.append(t.clone())
                return t
        torch.rand, torch.randn, torch.randint, torch.FloatTensor = wrap(torch.rand), wrap(torch.randn), wrap(torch.randint), FT
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn, torch.randint, torch.FloatTensor = self.saved


class Loader:
    """what the reference's DataLoader hands sample_data_basket: (images [n_d * acml * B, 3, S, S], labels) per draw"""

    def __init__(self, baskets):
        self.baskets = baskets

    def __iter__(self):
        return iter(self.baskets)


def reference_worker(R, cfgs, Gr, Dr, baskets, aa_p, freeze_d=-1):
    """the reference's WORKER built by ITS constructor on the CPU (global_rank 1: no wandb session)"""
    R._prepare()
    W = importlib.import_module("worker")
    ema_mod = importlib.import_module("utils.ema")
    RUN = cfgs.RUN
    for k, v in dict(mixed_precision=False, distributed_data_parallel=False, synchronized_bn=False, freezeD=freeze_d, empty_cache=False, langevin_sampling=False,
                     batch_statistics=False, train=True, project="x", entity="x", save_dir="/tmp").items():
        setattr(RUN, k, v)
    cfgs.OPTIMIZATION.world_size = 1
    Gema, ema = None, None
    if cfgs.MODEL.apply_g_ema:
        Gema = copy.deepcopy(Gr)
        ema = ema_mod.Ema(source=Gr, target=Gema, decay=cfgs.MODEL.g_ema_decay, start_iter=cfgs.MODEL.g_ema_start)
    cfgs.define_optimizer(Gr, Dr)
    w = W.WORKER(cfgs=cfgs, run_name="sweep", Gen=Gr, Gen_mapping=None, Gen_synthesis=None, Dis=Dr, Gen_ema=Gema, Gen_ema_mapping=None, Gen_ema_synthesis=None,
                 ema=ema, eval_model=None, train_dataloader=Loader(baskets), eval_dataloader=None, global_rank=1, local_rank="cpu", mu=None, sigma=None, real_feats=None,
                 logger=None, aa_p=aa_p, best_step=0, best_fid=None, best_ckpt_path=None, lecam_emas=None, num_eval={}, loss_list_dict={}, metric_dict_during_train={})
    for name in ("cond_loss", "cond_loss_mi"):          # (the constructor hard-codes master_rank="cuda" for the contrastive losses' masks)
        if hasattr(w, name) and hasattr(getattr(w, name), "master_rank"):
            getattr(w, name).master_rank = "cpu"
    w.topk = cfgs.OPTIMIZATION.batch_size
    w.prepare_train_iter(0)
    return w, Gema


def main():
    import fullemu
    from oracle import ref_import as R
    import studiogan_amd  # noqa: F401
    from studiogan_amd import config_map as CM
    from studiogan_amd import worker as SW
    from config_parity_emulated import shrink
    data = next((a[6:] for a in sys.argv[1:] if a.startswith("--dir=")), "CIFAR10")
    batch = int(next((a[8:] for a in sys.argv[1:] if a.startswith("--batch=")), "4"))
    only = set(a for a in sys.argv[1:] if not a.startswith("--"))
    global N_D
    N_D = int(next((a[5:] for a in sys.argv[1:] if a.startswith("--nd=")), str(N_D)))
    seed = int(next((a[7:] for a in sys.argv[1:] if a.startswith("--seed=")), "77"))
    n_steps = int(next((a[8:] for a in sys.argv[1:] if a.startswith("--steps=")), "1"))      # consecutive steps (the comparison is made after the last one)
    freeze_d = int(next((a[10:] for a in sys.argv[1:] if a.startswith("--freezeD=")), "-1"))      # RUN.freezeD: the first N discriminator blocks frozen (src/utils/misc.py:190-216)
    emit = next((a[7:] for a in sys.argv[1:] if a.startswith("--emit=")), None)      # write tests/golden/<emit>.npz / .json: draws + losses + gradient norms per file
    emitted, emit_meta = {}, {}
    acml = int(next((a[7:] for a in sys.argv[1:] if a.startswith("--acml=")), "1"))          # gradient accumulation (OPTIMIZATION.acml_steps): micro-batches per update
    files = sorted(glob.glob(f"/root/reference/src/configs/{data}/*.yaml"))
    torch.set_num_threads(1)
    dev = torch.device("cpu")
    n_ok = n_bad = 0
    worst_all = 0.0
    with fullemu.Installed(dma_late=1, greedy=1, seed=1):
        for f in files:
            name = os.path.basename(f)[:-5]
            y = yaml.safe_load(open(f))
            if "stylegan" in (y.get("MODEL") or {}).get("backbone", "resnet") or (only and name not in only):
                continue
            y = shrink(y)
            y.setdefault("OPTIMIZATION", {}).update(batch_size=batch, d_updates_per_step=N_D, acml_steps=acml)
            A = y.setdefault("AUG", {})
            if A.get("apply_ada"):
                A["ada_initial_augment_p"], A["ada_interval"] = 0.6, 1
            if A.get("apply_apa"):
                A["apa_initial_augment_p"], A["apa_interval"] = 0.5, 1
            if (y.get("LOSS") or {}).get("apply_lecam"):
                y["LOSS"]["lecam_ema_start_iter"] = 0          # (the files start the regulariser after 1000 steps: here it is live at step 1)
            t = time.time()
            try:
                torch.manual_seed(0)
                cfgs = R.load_cfgs({k: v for k, v in y.items() if k in ("DATA", "MODEL", "LOSS", "OPTIMIZATION", "AUG")})
                Gr, Dr = R.build_models(cfgs)
                g_state, d_state = copy.deepcopy(Gr.state_dict()), copy.deepcopy(Dr.state_dict())
                kw = CM.worker_kwargs(y)
                S, nc = (y.get("DATA") or {}).get("img_size", 32), kw["num_classes"]
                g = torch.Generator().manual_seed(11)
                nb = N_D * acml
                baskets = [(torch.randint(0, 256, (nb * batch, 3, S, S), generator=g).float() / 127.5 - 1.0, torch.randint(0, nc, (nb * batch,), generator=g)) for _ in range(n_steps * (1 + acml))]
                aa_p = A.get("ada_initial_augment_p", "N/A") if A.get("apply_ada") else A.get("apa_initial_augment_p", "N/A") if A.get("apply_apa") else "N/A"
                fm = kw["apply_fm"]
                per_step = 1 + (acml if fm else 0)          # baskets a step draws: one for its discriminator updates, one per micro-step of the feature-matching term
                # ---- the reference's worker
                rw, Gema_r = reference_worker(R, cfgs, Gr, Dr, baskets, aa_p, freeze_d)
                torch.manual_seed(seed)
                rec = RecordedDraws()
                with rec:
                    for step in range(1, n_steps + 1):
                        _, d_loss_r = rw.train_discriminator(step)
                        dg_r, dp_r = grads(Dr), params(Dr)
                        g_loss_r = rw.train_generator(step)
                        gg_r, gp_r = grads(Gr), params(Gr)
                aa_r = float(rw.aa_p) if aa_p != "N/A" else None
                # ---- this package's
                G, D, w = CM.build(y, dev)
                w.freezeD = freeze_d
                G.load_state_dict(g_state, strict=True)
                D.load_state_dict(d_state, strict=True)
                if w.Gen_ema is not None:
                    w.Gen_ema.load_state_dict(g_state, strict=True)
                torch.manual_seed(seed)
                for step in range(1, n_steps + 1):
                    b0 = (step - 1) * per_step
                    reals = [(baskets[b0][0][i * batch:(i + 1) * batch], baskets[b0][1][i * batch:(i + 1) * batch]) for i in range(nb)]
                    d_loss = w.train_discriminator(step, reals)
                    dg, dp = grads(D), params(D)
                    g_loss = w.train_generator(step, real_batches=[(baskets[b0 + 1 + i][0][:batch], baskets[b0 + 1 + i][1][:batch]) for i in range(acml)] if fm else None)
                    gg, gp = grads(G), params(G)
                d_loss, d_loss_r, g_loss, g_loss_r = d_loss.detach(), d_loss_r.detach(), g_loss.detach(), g_loss_r.detach()
                e_dl = abs(float(d_loss) - float(d_loss_r)) / max(abs(float(d_loss_r)), 1e-3)
                e_gl = abs(float(g_loss) - float(g_loss_r)) / max(abs(float(g_loss_r)), 1e-3)
                if kw["info_type"] != "N/A":
                    # InfoGAN's Q heads sit in the GENERATOR's optimiser (src/config.py:499-517): in a discriminator update they are frozen and their .grad is whatever the
                    # last generator update left (the reference) or zero (this package's fused discriminator Adam walks over them with exact zeros) -- not a gradient of
                    # this update on either side
                    for k in [k for k in dg_r if k.startswith(("info_discrete_linear", "info_conti_mu_linear", "info_conti_var_linear"))]:
                        dg_r.pop(k), dg.pop(k)
                e_dg, w_dg = worst(dg, dg_r)
                e_gg, w_gg = worst(gg, gg_r)
                if "--verbose" in sys.argv[1:]:
                    for tag, a, b in (("D", dg, dg_r), ("G", gg, gg_r)):
                        top = max(float(v.abs().max()) for v in b.values())
                        rows = sorted(((rel(a[k], b[k], floor=1e-2 * top), float((a[k] - b[k]).norm() / max(float(b[k].norm()), 1e-30)), float(b[k].abs().max()) / top, k) for k in b), reverse=True)[:6]
                        for r in rows:
                            print(f"    {tag} grad  max-rel {r[0]:.2e}  l2-rel {r[1]:.2e}  |g|/top {r[2]:.1e}  {r[3]}")
                lr_d, lr_g = kw["d_lr"], kw["g_lr"]
                # parameters after Adam: an element whose gradient is rounding noise moves by about +-lr per step with a sign two correct implementations need not share
                # (printed in units of lr per step; bound 3: Adam's bias-corrected step can exceed lr in its first steps)
                e_dp = max(float((dp[k] - dp_r[k]).abs().max()) for k in dp_r) / (N_D * n_steps * lr_d)
                e_gp = max(float((gp[k] - gp_r[k]).abs().max()) for k in gp_r) / (n_steps * lr_g)
                aa_txt, aa_ok, ema_ok, e_emb = "", True, True, 0.0
                if Gema_r is not None and w.Gen_ema is not None:          # the EMA twin (reference src/utils/ema.py:27-40): parameters in units of lr per step, buffers relative
                    pe, pr = dict(w.Gen_ema.named_parameters()), dict(Gema_r.named_parameters())
                    e_ema = max(float((pe[k].detach() - pr[k].detach()).abs().max()) for k in pr) / (n_steps * lr_g)
                    be, br = dict(w.Gen_ema.named_buffers()), dict(Gema_r.named_buffers())
                    e_emb = max([rel(be[k].float(), br[k].float(), floor=1e-3) for k in br if "num_batches" not in k and k in be] + [0.0])
                    nb_ok = all(int(be[k]) == int(br[k]) for k in br if "num_batches" in k and k in be)
                    aa_ok = nb_ok
                    ema_ok = e_ema <= 3.0 and e_emb <= 1e-3
                    aa_txt = f"  EMA {e_ema:.2f} lr, buffers {e_emb:.1e}"
                if aa_r is not None:
                    aa_mine = float(w.aa_p)
                    aa_ok = aa_ok and abs(aa_mine - aa_r) <= 1e-6
                    aa_txt += f"  aa_p {aa_mine:.6f} / {aa_r:.6f}"
                ok_first = e_dl <= 2e-3 and e_gl <= 2e-3 and e_dg <= 1e-2 and e_gg <= 1e-2 and e_dp <= 3.0 and e_gp <= 3.0 and aa_ok and ema_ok
                good, cond_txt = ok_first, ""
                if not ok_first and aa_ok and e_dl <= 5e-2 and e_gl <= 5e-2:
                    # losses agree, gradients do not: is THIS input ill-conditioned (a ReLU pre-activation within rounding distance of zero in a small early layer)? The
                    # reference's own movement under a perturbation of its initial weights of the size of the two implementations' forward discrepancy (their activations
                    # differ by ~3e-6 of the range: accumulation order; perturbation 2e-6 relative, as tests/golden/*.cond.npz) answers it: at a regular point the gradients
                    # move by ~1e-5, next to a tie (about 3e5 ReLU units per step at these sizes, each within 3e-6 of zero with probability ~1e-6: one step in three or
                    # four has one) by as much as two correct implementations differ -- sparsely when the unit sits in a 32 x 32 layer, densely in a 4 x 4 one
                    n_dg = n_gg = n_dl = n_gl = 0.0
                    for trial in range(8):          # (one perturbation lands on the other side of a tie about every second or third time)
                        torch.manual_seed(0)
                        cfgs2 = R.load_cfgs({k: v for k, v in y.items() if k in ("DATA", "MODEL", "LOSS", "OPTIMIZATION", "AUG")})
                        G2, D2 = R.build_models(cfgs2)
                        G2.load_state_dict(g_state, strict=True)
                        D2.load_state_dict(d_state, strict=True)
                        gp_ = torch.Generator().manual_seed(5 + trial)
                        with torch.no_grad():
                            for prm in list(G2.parameters()) + list(D2.parameters()):
                                prm.mul_(1.0 + 2e-6 * torch.randn(prm.shape, generator=gp_))
                        rw2, _ = reference_worker(R, cfgs2, G2, D2, baskets, aa_p, freeze_d)
                        torch.manual_seed(seed)
                        for st in range(1, n_steps + 1):
                            _, dl2 = rw2.train_discriminator(st)
                            if st == n_steps:
                                n_dl = max(n_dl, abs(float(dl2.detach()) - float(d_loss_r)) / max(abs(float(d_loss_r)), 1e-3))
                                n_dg = max(n_dg, worst({k: v for k, v in grads(D2).items() if k in dg_r}, dg_r)[0])
                            gl2 = rw2.train_generator(st)
                        n_gl = max(n_gl, abs(float(gl2.detach()) - float(g_loss_r)) / max(abs(float(g_loss_r)), 1e-3))
                        n_gg = max(n_gg, worst(grads(G2), gg_r)[0])
                    good = e_dg <= max(1e-2, 3 * n_dg) and e_gg <= max(1e-2, 3 * n_gg) and e_dl <= max(2e-3, 3 * n_dl) and e_gl <= max(2e-3, 3 * n_gl) and e_emb <= 1e-2      # (after a near-tie the twins' running statistics drift apart with the weights)
                    cond_txt = f"  [ill-conditioned input: the reference's own gradients move by D {n_dg:.1e} / G {n_gg:.1e} (its losses by {n_dl:.1e} / {n_gl:.1e}) under 2e-6 perturbations of its weights (worst of 8)]"
                n_ok += good
                n_bad += not good
                worst_all = max(worst_all, e_dl, e_gl, e_dg, e_gg)
                if emit and good:
                    for i, dr in enumerate(rec.draws):
                        emitted[f"{name}/draw{i}"] = dr.numpy()
                    emitted[f"{name}/d_loss"], emitted[f"{name}/g_loss"] = d_loss_r.cpu().numpy(), g_loss_r.cpu().numpy()
                    emitted[f"{name}/d_grad_norm"] = torch.stack([v.double().norm() for v in dg_r.values()]).norm().numpy()
                    emitted[f"{name}/g_grad_norm"] = torch.stack([v.double().norm() for v in gg_r.values()]).norm().numpy()
                    emit_meta[name] = {"yaml": {k: v for k, v in y.items() if k in ("DATA", "MODEL", "LOSS", "OPTIMIZATION", "AUG")}, "ill_conditioned": bool(cond_txt), "draws": len(rec.draws)}
                flags = [k[6:] for k, v in kw.items() if k.startswith("apply_") and v and k != "apply_g_ema"] + ([f"info:{kw['info_type']}"] if kw["info_type"] != "N/A" else [])
                print(f"{name:26s} {kw['adv_loss']:12s} {kw['d_cond_mtd']:6s} {kw['aux_cls_type']:4s} {','.join(flags):24s} | D loss {float(d_loss):+.5e} ({e_dl:.1e})  D grads {e_dg:.1e}  "
                      f"D params {e_dp:.2f} lr  | G loss {float(g_loss):+.5e} ({e_gl:.1e})  G grads {e_gg:.1e}  G params {e_gp:.2f} lr{aa_txt}{cond_txt}  "
                      f"{'ok' if good else 'MISMATCH ' + w_dg + ' / ' + w_gg} {time.time() - t:5.1f} s")
            except Exception as e:      # noqa: BLE001
                n_bad += 1
                import traceback
                tb = traceback.extract_tb(e.__traceback__)[-1]
                print(f"{name:26s} FAILED {type(e).__name__}: {str(e)[:240]}  [{os.path.basename(tb.filename)}:{tb.lineno}]")
            sys.stdout.flush()
    if emit:
        import json
        import numpy as np
        out = emit if os.sep in emit else os.path.join(ROOT, "tests", "golden", emit)
        np.savez_compressed(out + ".npz", **emitted)
        json.dump({"dir": data, "batch": batch, "n_d": N_D, "seed": seed, "cases": emit_meta}, open(out + ".json", "w"), indent=1)
        print(f"# wrote {out}.npz ({os.path.getsize(out + '.npz') // 1024} KiB), {len(emit_meta)} files")
    print(f"# {data}: {n_ok} configuration files: this package's training step agrees with the reference's own WORKER.train_discriminator / train_generator run on the CPU "
          f"(losses <= 2e-3, gradients <= 1e-2 of the largest; worst {worst_all:.1e}), {n_bad} do not")


if __name__ == "__main__":
    main()
