"""CPU oracle for the FID InceptionV3 feature extractor (reference src/metrics/inception_net.py:16-249).

TEST INFRASTRUCTURE ONLY (see oracle/restate.py header).

Pin: the reference's OWN code (fid_inception_v3, FIDInceptionA / C / E_1 / E_2.forward, InceptionV3.__init__ / forward) runs on CPU
with oracle/tv_stub.py standing in for the absent torchvision and this file's seeded weights as the "downloaded" checkpoint;
tests/test_oracle_cpu.py::test_inception_oracle_pinned_to_the_references_own_code asserts bit-identity block by block and end to end
at 299 x 299. PARITY UNPINNED only for the VALUES of the pretrained checkpoint: the reference downloads
`pt_inception-2015-12-05-6726825d.pth` at run time (inception_net.py:13,130) -- unavailable offline -- and torchvision itself (docker pin
torch 1.13 / torchvision 0.14) is restated from its published structure (BasicConv2d = Conv2d(bias=False) -> BatchNorm2d(eps=1e-3) ->
ReLU; Inception A/B/C/D/E), not executed. This file restates that structure with the FID patches of inception_net.py:135-249 (average
pools with count_include_pad=False in A, C, E_1; max pool in E_2; fc 2048 -> 1008), keyed by torchvision's state_dict names so the
real FID weights load.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3

# (name, cin, cout, (kh, kw), stride, (ph, pw)) for every BasicConv2d, in forward order per block
def _A(pre, cin, pf):
    return [(pre + ".branch1x1", cin, 64, (1, 1), 1, (0, 0)),
            (pre + ".branch5x5_1", cin, 48, (1, 1), 1, (0, 0)), (pre + ".branch5x5_2", 48, 64, (5, 5), 1, (2, 2)),
            (pre + ".branch3x3dbl_1", cin, 64, (1, 1), 1, (0, 0)), (pre + ".branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)),
            (pre + ".branch3x3dbl_3", 96, 96, (3, 3), 1, (1, 1)), (pre + ".branch_pool", cin, pf, (1, 1), 1, (0, 0))]


def _B(pre, cin):
    return [(pre + ".branch3x3", cin, 384, (3, 3), 2, (0, 0)), (pre + ".branch3x3dbl_1", cin, 64, (1, 1), 1, (0, 0)),
            (pre + ".branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)), (pre + ".branch3x3dbl_3", 96, 96, (3, 3), 2, (0, 0))]


def _C(pre, cin, c7):
    return [(pre + ".branch1x1", cin, 192, (1, 1), 1, (0, 0)),
            (pre + ".branch7x7_1", cin, c7, (1, 1), 1, (0, 0)), (pre + ".branch7x7_2", c7, c7, (1, 7), 1, (0, 3)),
            (pre + ".branch7x7_3", c7, 192, (7, 1), 1, (3, 0)),
            (pre + ".branch7x7dbl_1", cin, c7, (1, 1), 1, (0, 0)), (pre + ".branch7x7dbl_2", c7, c7, (7, 1), 1, (3, 0)),
            (pre + ".branch7x7dbl_3", c7, c7, (1, 7), 1, (0, 3)), (pre + ".branch7x7dbl_4", c7, c7, (7, 1), 1, (3, 0)),
            (pre + ".branch7x7dbl_5", c7, 192, (1, 7), 1, (0, 3)), (pre + ".branch_pool", cin, 192, (1, 1), 1, (0, 0))]


def _D(pre, cin):
    return [(pre + ".branch3x3_1", cin, 192, (1, 1), 1, (0, 0)), (pre + ".branch3x3_2", 192, 320, (3, 3), 2, (0, 0)),
            (pre + ".branch7x7x3_1", cin, 192, (1, 1), 1, (0, 0)), (pre + ".branch7x7x3_2", 192, 192, (1, 7), 1, (0, 3)),
            (pre + ".branch7x7x3_3", 192, 192, (7, 1), 1, (3, 0)), (pre + ".branch7x7x3_4", 192, 192, (3, 3), 2, (0, 0))]


def _E(pre, cin):
    return [(pre + ".branch1x1", cin, 320, (1, 1), 1, (0, 0)), (pre + ".branch3x3_1", cin, 384, (1, 1), 1, (0, 0)),
            (pre + ".branch3x3_2a", 384, 384, (1, 3), 1, (0, 1)), (pre + ".branch3x3_2b", 384, 384, (3, 1), 1, (1, 0)),
            (pre + ".branch3x3dbl_1", cin, 448, (1, 1), 1, (0, 0)), (pre + ".branch3x3dbl_2", 448, 384, (3, 3), 1, (1, 1)),
            (pre + ".branch3x3dbl_3a", 384, 384, (1, 3), 1, (0, 1)), (pre + ".branch3x3dbl_3b", 384, 384, (3, 1), 1, (1, 0)),
            (pre + ".branch_pool", cin, 192, (1, 1), 1, (0, 0))]


STEM = [("Conv2d_1a_3x3", 3, 32, (3, 3), 2, (0, 0)), ("Conv2d_2a_3x3", 32, 32, (3, 3), 1, (0, 0)), ("Conv2d_2b_3x3", 32, 64, (3, 3), 1, (1, 1)),
        ("Conv2d_3b_1x1", 64, 80, (1, 1), 1, (0, 0)), ("Conv2d_4a_3x3", 80, 192, (3, 3), 1, (0, 0))]
LAYERS = (STEM + _A("Mixed_5b", 192, 32) + _A("Mixed_5c", 256, 64) + _A("Mixed_5d", 288, 64) + _B("Mixed_6a", 288) +
          _C("Mixed_6b", 768, 128) + _C("Mixed_6c", 768, 160) + _C("Mixed_6d", 768, 160) + _C("Mixed_6e", 768, 192) +
          _D("Mixed_7a", 768) + _E("Mixed_7b", 1280) + _E("Mixed_7c", 2048))
SPEC = {l[0]: l for l in LAYERS}


def random_state_dict(seed=0):
    """Seeded random weights under torchvision's state_dict names (94 BasicConv2d + fc)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, cin, cout, (kh, kw), _, _ in LAYERS:
        fan = cin * kh * kw
        sd[name + ".conv.weight"] = torch.randn(cout, cin, kh, kw, generator=g) * math.sqrt(2.0 / fan)
        sd[name + ".bn.weight"] = 1.0 + 0.1 * torch.randn(cout, generator=g)
        sd[name + ".bn.bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[name + ".bn.running_mean"] = 0.1 * torch.randn(cout, generator=g)
        sd[name + ".bn.running_var"] = 0.5 + torch.rand(cout, generator=g)
        sd[name + ".bn.num_batches_tracked"] = torch.tensor(0)
    sd["fc.weight"] = torch.randn(1008, 2048, generator=g) * math.sqrt(1.0 / 2048)
    sd["fc.bias"] = 0.01 * torch.randn(1008, generator=g)
    return sd


def _bc(x, sd, name):
    _, _, _, _, stride, pad = SPEC[name]
    x = F.conv2d(x, sd[name + ".conv.weight"], None, stride=stride, padding=pad)
    x = F.batch_norm(x, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"],
                     False, 0.1, BN_EPS)
    return F.relu(x)


def _blockA(x, sd, p):   # inception_net.py:135-156 (FIDInceptionA)
    b1 = _bc(x, sd, p + ".branch1x1")
    b5 = _bc(_bc(x, sd, p + ".branch5x5_1"), sd, p + ".branch5x5_2")
    b3 = _bc(_bc(_bc(x, sd, p + ".branch3x3dbl_1"), sd, p + ".branch3x3dbl_2"), sd, p + ".branch3x3dbl_3")
    bp = _bc(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False), sd, p + ".branch_pool")
    return torch.cat([b1, b5, b3, bp], 1)


def _blockB(x, sd, p):   # torchvision InceptionB (unpatched)
    b3 = _bc(x, sd, p + ".branch3x3")
    bd = _bc(_bc(_bc(x, sd, p + ".branch3x3dbl_1"), sd, p + ".branch3x3dbl_2"), sd, p + ".branch3x3dbl_3")
    return torch.cat([b3, bd, F.max_pool2d(x, 3, 2)], 1)


def _blockC(x, sd, p):   # inception_net.py:159-183 (FIDInceptionC)
    b1 = _bc(x, sd, p + ".branch1x1")
    b7 = _bc(_bc(_bc(x, sd, p + ".branch7x7_1"), sd, p + ".branch7x7_2"), sd, p + ".branch7x7_3")
    bd = x
    for i in range(1, 6):
        bd = _bc(bd, sd, p + f".branch7x7dbl_{i}")
    bp = _bc(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False), sd, p + ".branch_pool")
    return torch.cat([b1, b7, bd, bp], 1)


def _blockD(x, sd, p):   # torchvision InceptionD (unpatched)
    b3 = _bc(_bc(x, sd, p + ".branch3x3_1"), sd, p + ".branch3x3_2")
    b7 = x
    for i in range(1, 5):
        b7 = _bc(b7, sd, p + f".branch7x7x3_{i}")
    return torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)


def _blockE(x, sd, p, pool):   # inception_net.py:186-249 (FIDInceptionE_1: avg w/o pad count, E_2: max pool)
    b1 = _bc(x, sd, p + ".branch1x1")
    t = _bc(x, sd, p + ".branch3x3_1")
    b3 = torch.cat([_bc(t, sd, p + ".branch3x3_2a"), _bc(t, sd, p + ".branch3x3_2b")], 1)
    t = _bc(_bc(x, sd, p + ".branch3x3dbl_1"), sd, p + ".branch3x3dbl_2")
    bd = torch.cat([_bc(t, sd, p + ".branch3x3dbl_3a"), _bc(t, sd, p + ".branch3x3dbl_3b")], 1)
    pooled = F.avg_pool2d(x, 3, 1, 1, count_include_pad=False) if pool == "avg" else F.max_pool2d(x, 3, 1, 1)
    bp = _bc(pooled, sd, p + ".branch_pool")
    return torch.cat([b1, b3, bd, bp], 1)


def inception_forward(x, sd):
    """x: [B,3,299,299] in [-1,1] (resize_input=False, normalize_input=False: metrics/preparation.py:53).
    Returns (pool3 features [B,2048], logits [B,1008])  -- inception_net.py:81-107."""
    x = _bc(_bc(_bc(x, sd, "Conv2d_1a_3x3"), sd, "Conv2d_2a_3x3"), sd, "Conv2d_2b_3x3")
    x = F.max_pool2d(x, 3, 2)
    x = _bc(_bc(x, sd, "Conv2d_3b_1x1"), sd, "Conv2d_4a_3x3")
    x = F.max_pool2d(x, 3, 2)
    x = _blockA(x, sd, "Mixed_5b"); x = _blockA(x, sd, "Mixed_5c"); x = _blockA(x, sd, "Mixed_5d")
    x = _blockB(x, sd, "Mixed_6a")
    for n in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
        x = _blockC(x, sd, n)
    x = _blockD(x, sd, "Mixed_7a")
    x = _blockE(x, sd, "Mixed_7b", "avg")
    x = _blockE(x, sd, "Mixed_7c", "max")
    feat = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    return feat, F.linear(feat, sd["fc.weight"], sd["fc.bias"])


def quantize_resize_normalize(x, quantize=True, size=299):
    """ops.quantize_images + 'legacy' resizer + normalisation (utils/ops.py:251-263, utils/resize.py:72-93):
    uint8 truncation of (x+1)/2*255+0.5, per-image bilinear (align_corners=False) to 299, clip, (v/255-0.5)/0.5."""
    import numpy as np
    if quantize:
        q = (x + 1) / 2
        q = (255.0 * q + 0.5).clamp(0.0, 255.0)
        q = q.detach().cpu().numpy().astype(np.uint8)
    else:
        q = x.detach().cpu().numpy().astype(np.uint8)
    r = F.interpolate(torch.from_numpy(q).float(), size=(size, size), mode="bilinear", align_corners=False).clamp(0, 255)
    return (r / 255.0 - 0.5) / 0.5, q


# ---- IS / FID back-end restated (metrics/ins.py:28-42,62-76 ; metrics/fid.py:34-98) -------------------------------
def inception_score(probs, splits=1):
    """exp(mean_n sum_c p (log p - log mean p)) per split (ins.py:28-42)."""
    scores = []
    n = probs.shape[0]
    for j in range(splits):
        part = probs[(j * n // splits):((j + 1) * n // splits)]
        kl = part * (torch.log(part) - torch.log(torch.unsqueeze(torch.mean(part, 0), 0)))
        scores.append(torch.exp(torch.mean(torch.sum(kl, 1))))
    return torch.stack(scores)


def topk_hits(scores, labels, k):
    """sklearn.metrics.top_k_accuracy_score semantics (stable ascending argsort, reversed): the true class is a hit iff
    fewer than k classes beat it, where a tie is won by the HIGHER class index."""
    s_true = scores.gather(1, labels.view(-1, 1))
    idx = torch.arange(scores.shape[1]).view(1, -1)
    beat = (scores > s_true) | ((scores == s_true) & (idx > labels.view(-1, 1)))
    return beat.sum(1) < k


def frechet_distance(mu1, sigma1, mu2, sigma2):
    """fid.py:34-62 (scipy.linalg.sqrtm on the host, fp64)."""
    import numpy as np
    from scipy import linalg
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(sigma1.shape[0]) * 1e-6
        covmean = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def prdc(real, fake, nearest_k):
    """reference src/metrics/prdc.py:87-168 restated in float64 torch (euclidean pairwise distances; manifold radius = the
    (nearest_k + 1)-th smallest self-inclusive distance of a sample, i.e. get_kth_value(distances, k = nearest_k + 1)): returns
    dict(precision, recall, density, coverage). Pinned against the reference's compute_prdc by oracle/make_golden_metrics.py."""
    real, fake = real.double(), fake.double()

    def radii(x):
        d = torch.cdist(x, x)
        return torch.kthvalue(d, nearest_k + 1, dim=1).values

    rr, rf = radii(real), radii(fake)
    d = torch.cdist(real, fake)
    inside_real = d < rr.unsqueeze(1)
    precision = inside_real.any(dim=0).double().mean()
    recall = (d < rf.unsqueeze(0)).any(dim=1).double().mean()
    density = (1.0 / float(nearest_k)) * inside_real.sum(dim=0).double().mean()
    coverage = (d.min(dim=1).values < rr).double().mean()
    return dict(precision=float(precision), recall=float(recall), density=float(density), coverage=float(coverage))
