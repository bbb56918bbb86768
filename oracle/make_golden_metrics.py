"""Pin the host-side metric formulas of the product (pytorch-studiogan_amd/metrics.py: frechet_inception_distance, calculate_kl_div)
and of the oracle (oracle/inception.py: frechet_distance) against the REAL reference functions (reference src/metrics/fid.py:34-62,
src/metrics/ins.py:28-42) run here on CPU; writes tests/golden/metrics_host.npz (inputs + the reference's outputs).

    python oracle/make_golden_metrics.py      (authoring container only: imports /root/reference through oracle/ref_import.py)

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

OUT = os.path.join(ROOT, "tests", "golden", "metrics_host")


def main():
    RI._prepare()
    fid = importlib.import_module("metrics.fid")
    ins = importlib.import_module("metrics.ins")
    rs = np.random.RandomState(20260922)
    fix = {}
    # FID between two Gaussians' sample moments (64-d), and a near-singular pair (rank-deficient covariance: the eps branch of fid.py:50-54)
    a = rs.randn(400, 64)
    b = rs.randn(400, 64) * 1.3 + 0.2
    c = np.concatenate([rs.randn(30, 64)] * 2)         # 60 samples, rank <= 29
    for tag, x in (("a", a), ("b", b), ("c", c)):
        fix[f"in/mu_{tag}"], fix[f"in/sigma_{tag}"] = x.mean(0), np.cov(x, rowvar=False)
    fix["exp/fid_ab"] = np.float64(fid.frechet_inception_distance(fix["in/mu_a"], fix["in/sigma_a"], fix["in/mu_b"], fix["in/sigma_b"]))
    fix["exp/fid_ac"] = np.float64(fid.frechet_inception_distance(fix["in/mu_a"], fix["in/sigma_a"], fix["in/mu_c"], fix["in/sigma_c"]))
    fix["exp/fid_aa"] = np.float64(fid.frechet_inception_distance(fix["in/mu_a"], fix["in/sigma_a"], fix["in/mu_a"], fix["in/sigma_a"]))
    # Inception score of softmax rows, 1 and 5 splits (ins.py:28-42)
    p = torch.softmax(3.0 * torch.randn(250, 1008, generator=torch.Generator().manual_seed(5)), 1)
    fix["in/probs"] = p.numpy()
    for splits in (1, 5):
        m, s = ins.calculate_kl_div(p, splits)
        fix[f"exp/is_mean_{splits}"], fix[f"exp/is_std_{splits}"] = np.float64(m), np.float64(s)
    # Evaluation pre-processing (src/metrics/preparation.py:103-108 -> utils/ops.py:251-263 quantize_images + resize_images with the
    # "legacy" resizer of utils/resize.py:68-69,83-93). torchvision is absent here: ToTensor is applied by its definition for a float
    # HWC ndarray (transpose to CHW, no scaling -- only uint8 input is divided by 255).
    rops = importlib.import_module("utils.ops")
    rres = importlib.import_module("utils.resize")
    from oracle import inception as OI
    x = torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(9)) * 2.4 - 1.2          # incl. values outside [-1, 1]
    q = rops.quantize_images(x)
    resizer = rres.build_resizer("legacy", "InceptionV3_tf", 299)
    to_tensor = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose((2, 0, 1))))
    mean = torch.Tensor([0.5, 0.5, 0.5]).view(1, 3, 1, 1)
    std = torch.Tensor([0.5, 0.5, 0.5]).view(1, 3, 1, 1)
    r = rops.resize_images(q, resizer, to_tensor, mean, std, device="cpu")
    ro, qo = OI.quantize_resize_normalize(x, quantize=True)
    assert np.array_equal(q, qo), "uint8 quantisation"
    assert torch.equal(r, ro), "legacy resize + normalise"
    flat = r.reshape(-1)
    idx = (torch.arange(4096, dtype=torch.int64) * flat.numel()) // 4096
    fix["in/pre_x"], fix["exp/pre_q"] = x.numpy(), q
    fix["exp/pre_samples"], fix["exp/pre_norms"] = flat[idx].numpy(), np.array([float(flat.double().sum()), float(flat.double().norm())])
    print("eval pre-processing: oracle bit-identical to the reference (uint8 image and the 299x299 normalised tensor)")
    # FID moments of generated features: the reference truncates the (over-generated) stack to num_generate BEFORE np.mean / np.cov
    # (src/metrics/fid.py:64-69,96-98); 37 rows = 5 batches of 8 minus nothing, num_generate = 30 is not a multiple of the batch
    ff = torch.randn(37, 32, generator=torch.Generator().manual_seed(12)) * 1.7 + 0.3
    mu, sigma = fid.calculate_moments(data_loader="N/A", eval_model=None, num_generate=30, batch_size=8, quantize=True, world_size=1, DDP=False,
                                      disable_tqdm=True, fake_feats=ff)
    fix["in/mom_feats"], fix["in/mom_num_generate"] = ff.numpy(), np.int64(30)
    fix["exp/mom_mu"], fix["exp/mom_sigma"] = mu.astype(np.float64), sigma.astype(np.float64)
    # PRDC (src/metrics/prdc.py:143-168): two overlapping Gaussian clouds, nearest_k = 5 (the reference's default)
    prdc = importlib.import_module("metrics.prdc")
    pr = torch.randn(300, 64, generator=torch.Generator().manual_seed(21)).numpy().astype(np.float64)
    pf = (torch.randn(260, 64, generator=torch.Generator().manual_seed(22)) * 1.15 + 0.35).numpy().astype(np.float64)
    m = prdc.compute_prdc(real_features=pr, fake_features=pf, nearest_k=5)
    mo = OI.prdc(torch.from_numpy(pr), torch.from_numpy(pf), 5)
    for k in ("precision", "recall", "density", "coverage"):
        assert abs(float(m[k]) - mo[k]) < 1e-12, ("PRDC restatement", k, float(m[k]), mo[k])
        fix["exp/prdc_" + k] = np.float64(m[k])
    fix["in/prdc_real"], fix["in/prdc_fake"] = pr.astype(np.float32), pf.astype(np.float32)
    print("PRDC: oracle restatement identical to the reference's compute_prdc:", {k: float(v) for k, v in m.items()})
    np.savez_compressed(OUT + ".npz", **fix)
    json.dump({"note": "reference src/metrics/fid.py frechet_inception_distance and src/metrics/ins.py calculate_kl_div run on CPU by "
                       "oracle/make_golden_metrics.py"}, open(OUT + ".json", "w"), indent=1)
    print({k: float(v) for k, v in fix.items() if k.startswith("exp/") and np.ndim(v) == 0})
    print("wrote", OUT + ".npz", os.path.getsize(OUT + ".npz") // 1024, "KiB")


if __name__ == "__main__":
    main()
