"""Golden vectors of the evaluation-side generator preparation written by the REAL reference (reference src/utils/misc.py:63-107 GeneratorController.prepare_generator,
:301-334 apply_standing_statistics, as src/worker.py:815-816 calls them in front of the FID / IS feature extraction): for three width-8 generators (conditional batch
norm ResNet; BigGAN with spectral norm, shared embedding and attention; unconditional ResNet with plain batch norm) the running statistics after `standing_step` training-mode forwards
over batches of random sizes, and the evaluation-mode image of fixed latents in each of the three modes (standing statistics / batch statistics / plain eval). The draws
(labels, latents) are recorded in call order so that the product replays them. No restatement: the product is held against these vectors (tests/aug_checks.py
standing_case). Output: tests/golden/standing.npz (+ .json).

    python -m oracle.make_golden_standing           (authoring container only: needs /root/reference)
TEST INFRASTRUCTURE."""
import copy
import importlib
import json
import os
import random
import types

import numpy as np
import torch

from . import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "standing")
CASES = {
    "sngan": {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
              "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8}},
    "biggan": {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
               "MODEL": {"backbone": "big_resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True, "apply_attn": True,
                         "attn_g_loc": [2], "attn_d_loc": [1], "z_dim": 40, "g_shared_dim": 16, "g_conv_dim": 8, "d_conv_dim": 8}},
    "resgan": {"DATA": {"name": "CIFAR10", "img_size": 32, "num_classes": 10},
               "MODEL": {"backbone": "resnet", "apply_d_sn": True, "z_prior": "gaussian", "z_dim": 32, "g_conv_dim": 8, "d_conv_dim": 8}},
}
MAX_BATCH, STEPS, SEED = 7, 5, 4243


class Recorded:
    """torch.randint / torch.randn with their results recorded in call order"""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self.saved = (torch.randint, torch.randn)
        i0, n0, draws = torch.randint, torch.randn, self.draws

        def randint(*a, **k):
            t = i0(*a, **k)
            draws.append(t.clone())
            return t

        def randn(*a, **k):
            t = n0(*a, **k)
            draws.append(t.clone())
            return t
        torch.randint, torch.randn = randint, randn
        return self

    def __exit__(self, *a):
        torch.randint, torch.randn = self.saved


def main():
    assert RI.available(), "needs the reference checkout"
    RI._prepare()
    misc = importlib.import_module("utils.misc")
    fix, meta = {}, {"cases": {}, "max_batch": MAX_BATCH, "steps": STEPS, "seed": SEED}
    logger = types.SimpleNamespace(info=lambda *a, **k: None)
    for ci, (name, y) in enumerate(CASES.items()):
        cfgs = RI.load_cfgs(y)
        cfgs.OPTIMIZATION.world_size = 1
        cfgs.RUN.distributed_data_parallel = False
        cfgs.RUN.langevin_sampling = False
        torch.manual_seed(SEED + ci)
        Gen, _ = RI.build_models(cfgs)
        # a generator that has trained a little: non-trivial running statistics and spectral-norm vectors (two tracked training forwards)
        g = torch.Generator().manual_seed(SEED + 10 + ci)
        zd, nc = cfgs.MODEL.z_dim, cfgs.DATA.num_classes
        with torch.no_grad():
            for _ in range(2):
                Gen(torch.randn(4, zd, generator=g), torch.randint(0, nc, (4,), generator=g))
        pre = name + "/"
        for k, v in Gen.state_dict().items():
            fix[pre + "init/" + k] = v.clone()
        z_eval, y_eval = torch.randn(3, zd, generator=g), torch.randint(0, nc, (3,), generator=g)
        fix[pre + "z_eval"], fix[pre + "y_eval"] = z_eval, y_eval
        sizes = {}
        for mode in ("standing", "batch", "plain"):
            G = copy.deepcopy(Gen)
            ctl = misc.GeneratorController(generator=G, generator_mapping=None, generator_synthesis=None, batch_statistics=mode == "batch",
                                           standing_statistics=mode == "standing", standing_max_batch=MAX_BATCH, standing_step=STEPS, cfgs=cfgs, device="cpu",
                                           global_rank=1, logger=logger, std_stat_counter=0)
            random.seed(SEED + 20 + ci)
            torch.manual_seed(SEED + 30 + ci)
            with Recorded() as rec, torch.no_grad():
                G, _, _ = ctl.prepare_generator()
            if mode == "standing":
                for i, d in enumerate(rec.draws):
                    fix[pre + f"draw{i}"] = d
                sizes = [int(d.shape[0]) for d in rec.draws[::2]]
                assert len(rec.draws) == 2 * STEPS
            with torch.no_grad():
                img = G(z_eval, y_eval, eval=True)
            fix[pre + mode + "/image"] = img.clone()
            for k, v in G.named_buffers():          # (the parameters do not move; batch-norm statistics and spectral-norm vectors do)
                fix[pre + mode + "/final/" + k] = v.clone()
            flags = sorted({(type(m).__name__, m.training) for m in G.modules() if not list(m.children())})
            meta["cases"].setdefault(name, {"yaml": y})[mode + "_training_flags"] = [[a, bool(b)] for a, b in flags]
        meta["cases"][name]["batch_sizes"] = sizes
        print(f"{name}: standing batches {sizes}; image |max| standing {float(fix[pre + 'standing/image'].abs().max()):.4f} "
              f"batch {float(fix[pre + 'batch/image'].abs().max()):.4f} plain {float(fix[pre + 'plain/image'].abs().max()):.4f}")
    np.savez_compressed(OUT + ".npz", **{k: v.detach().cpu().numpy() for k, v in fix.items()})
    json.dump(meta, open(OUT + ".json", "w"), indent=1)
    print(f"wrote {OUT}.npz {os.path.getsize(OUT + '.npz') // 1024} KiB")


if __name__ == "__main__":
    main()
