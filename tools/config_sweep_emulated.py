"""One training step (d_updates_per_step capped at 1 discriminator update + one generator update) of EVERY non-StyleGAN CIFAR10 configuration file of the reference,
built at width 8 through studiogan_amd.config_map and run through the kernel SOURCES on the CPU interpreter (tests/hipemu): does every configuration's combination of
heads / losses / regularisers / augmentations execute end to end with finite losses? (Parity of each ingredient is the business of the golden-vector tests; this is
the integration sweep.)   usage: python tools/config_sweep_emulated.py [--bf16] [--dir=ImageNet] [--batch=2] [name ...] > profiles/<...>.txt      TEST INFRASTRUCTURE; needs /root/reference for the files."""
import glob
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))


def main():
    import fullemu
    import studiogan_amd  # noqa: F401
    from studiogan_amd import config_map as CM
    mixed = "--bf16" in sys.argv[1:]
    data = next((a[6:] for a in sys.argv[1:] if a.startswith("--dir=")), "CIFAR10")
    batch = int(next((a[8:] for a in sys.argv[1:] if a.startswith("--batch=")), "4"))
    only = set(a for a in sys.argv[1:] if not a.startswith("--"))
    files = sorted(glob.glob(f"/root/reference/src/configs/{data}/*.yaml"))
    torch.set_num_threads(1)
    dev = torch.device("cpu")
    ok = bad = 0
    with fullemu.Installed(dma_late=1, greedy=1, seed=1):
        for f in files:
            name = os.path.basename(f)[:-5]
            y = yaml.safe_load(open(f))
            if "stylegan" in (y.get("MODEL") or {}).get("backbone", "resnet") or (only and name not in only):
                continue
            y.setdefault("MODEL", {})
            for k in ("g_conv_dim", "d_conv_dim"):
                if y["MODEL"].get(k, 64) != "N/A":
                    y["MODEL"][k] = 8
            if y["MODEL"].get("d_embed_dim", "N/A") != "N/A":
                y["MODEL"]["d_embed_dim"] = 16
            if y["MODEL"].get("g_shared_dim", "N/A") != "N/A":
                y["MODEL"]["g_shared_dim"] = 16
            y.setdefault("OPTIMIZATION", {})["batch_size"] = batch
            y["OPTIMIZATION"]["d_updates_per_step"] = 1
            y["OPTIMIZATION"]["acml_steps"] = 1
            t = time.time()
            try:
                torch.manual_seed(0)
                bb = CM.model_args(y)[0]
                G, D, w = CM.build(y, dev, mixed_precision=mixed)
                kw = CM.worker_kwargs(y)
                B = kw["batch_size"]
                S = (y.get("DATA") or {}).get("img_size", 32)
                real = (torch.randint(0, 256, (B, 3, S, S)).float() / 127.5 - 1.0, torch.randint(0, kw["num_classes"], (B,)))
                d, g = w.step(0, [real])
                fin = bool(torch.isfinite(d)) and bool(torch.isfinite(g)) and all(bool(torch.isfinite(p).all()) for p in list(G.parameters()) + list(D.parameters()))
                flags = [k[6:] for k, v in kw.items() if k.startswith("apply_") and v and k != "apply_g_ema"] + ([f"info:{kw['info_type']}"] if kw["info_type"] != "N/A" else [])
                print(f"{name:24s} {bb:26s} {kw['adv_loss']:12s} d_cond {kw['d_cond_mtd']:6s} aux {kw['aux_cls_type']:4s} {','.join(flags):28s} d_loss {float(d):+.4e} g_loss {float(g):+.4e} "
                      f"{'finite' if fin else 'NON-FINITE'} {time.time() - t:5.1f} s")
                ok += fin
                bad += not fin
            except Exception as e:      # noqa: BLE001
                bad += 1
                print(f"{name:24s} FAILED {type(e).__name__}: {str(e)[:300]}")
            sys.stdout.flush()
    print(f"# {'bf16' if mixed else 'fp32'}: {ok} configurations ran one step with finite results, {bad} did not")


if __name__ == "__main__":
    main()
