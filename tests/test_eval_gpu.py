"""GPU: FID/IS feature-extraction path (studiogan_amd/metrics.py) against the CPU oracle (oracle/inception.py) with the
same seeded random Inception weights (pretrained weights are unavailable offline: parity is structural, see the oracle
header), plus the integer parts that must be bit-exact (uint8 quantisation, top-k with sklearn's tie rule)."""
import numpy as np
import pytest
import torch

from util import check
from oracle import inception as OI

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_inception_features_vs_oracle(sg, dtype, tol):
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    sd = OI.random_state_dict(0)
    g = torch.Generator().manual_seed(5)
    imgs = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    x299, q = OI.quantize_resize_normalize(imgs, quantize=True)
    feat_o, logit_o = OI.inception_forward(x299, sd)
    model = M.LoadEvalModel("InceptionV3_tf", "legacy", 1, False, dev, state_dict=sd, dtype=dtype)
    feat, logit = model.get_outputs(imgs.to(dev), quantize=True)
    torch.cuda.synchronize()
    check(f"inception pool3 features {dtype}", feat, feat_o, tol)
    check(f"inception logits {dtype}", logit, logit_o, tol)
    p = M.softmax_rows(logit)
    check("softmax(logits)", p, torch.softmax(logit.cpu().double(), 1), 1e-5)


def test_inception_features_at_full_input_size(sg):
    """InceptionV3 at the real evaluation geometry: 32 images already at 299 x 299 (no resize in the way), fp32 <= 2e-4 against
    the CPU oracle (itself bit-identical to the reference's inception_net.py, tests/test_oracle_cpu.py); bf16 reported with its bound."""
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    sd = OI.random_state_dict(1)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(32, 3, 299, 299, generator=g) * 2 - 1
    feat_o, logit_o = OI.inception_forward(x, sd)
    for dtype, tol in ((torch.float32, 2e-4), (torch.bfloat16, 6e-2)):
        model = M.InceptionV3(sd, dev, dtype)
        xn = x.to(dev).permute(0, 2, 3, 1).contiguous().to(dtype)
        feat, logit = model.forward_nhwc(xn)
        torch.cuda.synchronize()
        check(f"299^2 B=32 pool3 features {dtype}", feat, feat_o, tol)
        check(f"299^2 B=32 logits {dtype}", logit, logit_o, tol)


def test_inception_features_bf16x3_split_mode(sg):
    """InceptionV3 in fp32 with every convolution in the "bf16x3" arithmetic (fp32 tensors, operands split into two bf16 terms in registers, three bf16 MFMAs per
    k-tile: functional.f32_mode, csrc/gemm_core.h SPLIT) at the real evaluation geometry: the same 2e-4 bound against the CPU oracle as the exact fp32 path --
    measured 4e-6 (features) / 3e-6 (logits), i.e. 50x inside it and ~100x finer than TF32 -- and the mode must actually have run (results differ from the exact
    path's by more than fp32 rounding, the library's switch is back at "exact" afterwards)."""
    from studiogan_amd import metrics as M, _lib as L
    dev = torch.device("cuda:0")
    sd = OI.random_state_dict(1)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(32, 3, 299, 299, generator=g) * 2 - 1
    feat_o, logit_o = OI.inception_forward(x, sd)
    xn = x.to(dev).permute(0, 2, 3, 1).contiguous()
    out = {}
    for mode in ("exact", "bf16x3"):
        model = M.InceptionV3(sd, dev, torch.float32, f32_mode=mode)
        out[mode] = model.forward_nhwc(xn)
        torch.cuda.synchronize()
        assert L.lib().sg_get_f32_mode() == 0
    check("299^2 B=32 pool3 features fp32 bf16x3", out["bf16x3"][0], feat_o, 2e-4)
    check("299^2 B=32 logits fp32 bf16x3", out["bf16x3"][1], logit_o, 2e-4)
    d = float((out["bf16x3"][0] - out["exact"][0]).abs().max() / out["exact"][0].abs().max())
    assert 1e-7 < d < 1e-4, d


def test_preprocess_bit_exact_quantisation(sg):
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    imgs = torch.rand(3, 3, 32, 32, generator=g) * 2.4 - 1.2
    x_o, q_o = OI.quantize_resize_normalize(imgs, quantize=True)
    x, q = M.preprocess(imgs.to(dev), torch.float32, True, 299, want_uint8=True)
    assert np.array_equal(q.cpu().numpy(), q_o), "uint8 quantisation must be bit-exact"
    check("resize+normalise", x.cpu().permute(0, 3, 1, 2), x_o, 1e-5)


def test_topk_moments_is_fid(sg):
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    probs = torch.softmax(torch.randn(300, 1008, generator=g), 1)
    probs[:50, 1:1001] = probs[:50, 1:2]            # exact ties across all classes for the first rows
    probs[50:80, 5] = probs[50:80, 9]               # pairwise ties
    labels = torch.randint(0, 1000, (300,), generator=g)
    for k in (1, 5):
        ref = OI.topk_hits(probs[:, 1:1001], labels, k).double().mean().item()
        got = M.top_k_accuracy(probs.to(dev), labels, k, c0=1, ncls=1000)
        assert abs(ref - got) < 1e-6, (k, ref, got)
    # sklearn agrees with the restated tie rule
    from sklearn.metrics import top_k_accuracy_score
    sk = top_k_accuracy_score(labels.numpy(), probs[:, 1:1001].numpy(), k=5, labels=range(1000))
    assert abs(sk - OI.topk_hits(probs[:, 1:1001], labels, 5).double().mean().item()) < 1e-12
    # moments
    f = torch.randn(1000, 64, generator=g) * 2 + 0.5
    mu, sigma = M.calculate_moments(f.to(dev))
    check("moments mean", torch.from_numpy(mu), f.double().mean(0), 1e-6)
    check("moments cov", torch.from_numpy(sigma), torch.from_numpy(np.cov(f.double().numpy(), rowvar=False)), 1e-5)
    assert abs(M.frechet_inception_distance(mu, sigma, mu, sigma)) < 1e-6          # KAT: FID(x, x) = 0 (fid.py:34-62)
    u = torch.full((64, 1008), 1.0 / 1008, device=dev)
    m, _ = M.calculate_kl_div(u, 1)
    assert abs(float(m) - 1.0) < 1e-6                                                # KAT: IS of uniform predictions = 1


def test_feature_loop_with_generator(sg):
    """generate_images_and_stack_features end to end with a small BigGAN generator (reference features.py:17-65)."""
    from studiogan_amd import metrics as M
    from util import load_golden, sub
    from test_model_gpu import build_from_yaml
    dev = torch.device("cuda:0")
    fix, meta = load_golden("biggan32")
    G, _ = build_from_yaml(meta["yaml"], False, dev)
    G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
    G.eval()
    model = M.LoadEvalModel(device=dev, state_dict=OI.random_state_dict(0), dtype=torch.bfloat16)
    mom = M.FeatureMoments(2048, dev)
    feats, probs, labels = M.generate_images_and_stack_features(G, model, 10, 4, 40, 10, quantize=True, device=dev, moments=mom)
    assert feats.shape == (12, 2048) and probs.shape == (12, 1008) and len(labels) == 12
    assert torch.isfinite(feats).all() and abs(float(probs.sum(1).mean()) - 1) < 1e-4
    mu, sigma = mom.finalize()
    # the reference's rule (src/metrics/fid.py:68-69,96-97): moments over the first num_generate = 10 rows of the 12 generated
    assert mom.n == 10
    kept = feats[:10].double().cpu().numpy()
    check("loop moments mean (first num_generate rows)", torch.from_numpy(mu), torch.from_numpy(kept.mean(0)), 1e-5)
    check("loop moments cov (first num_generate rows)", torch.from_numpy(sigma), torch.from_numpy(np.cov(kept, rowvar=False)), 1e-4)


def test_moments_truncation_vs_reference_fixture(sg):
    """M.calculate_moments(feats, num_generate) against the REAL reference's calculate_moments(fake_feats=...) output for a stack whose
    length is not num_generate (tests/golden/metrics_host.npz, written by oracle/make_golden_metrics.py)."""
    import os
    from studiogan_amd import metrics as M
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_host.npz"))
    dev = torch.device("cuda:0")
    f = torch.from_numpy(z["in/mom_feats"]).to(dev)
    mu, sigma = M.calculate_moments(f, int(z["in/mom_num_generate"]))
    check("mu vs reference calculate_moments", torch.from_numpy(mu), torch.from_numpy(z["exp/mom_mu"]), 1e-6)
    check("sigma vs reference calculate_moments", torch.from_numpy(sigma), torch.from_numpy(z["exp/mom_sigma"]), 1e-5)


def test_topk_training_select_and_scatter(sg):
    """Top-k generator training (reference src/worker.py:565-566: torch.topk(adv_output, k).values, then the loss over them):
    the selected VALUES are bit-exact against torch.topk on the host (ties included), the gradient lands on the selected
    logits only, and k follows losses.adjust_k (reference src/utils/losses.py:364-366)."""
    from studiogan_amd import losses as SL
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    for n, k in ((256, 256), (256, 128), (64, 7), (1000, 333)):
        x = torch.randn(n, generator=g)
        x[n // 3] = x[n // 2]                                   # a tie
        xd = x.to(dev).requires_grad_(True)
        v = SL.topk_values(xd, k)
        ref = torch.topk(x, k).values
        assert torch.equal(v.detach().cpu(), ref), "top-k values must be bit-exact"
        w = torch.randn(k, generator=g)
        (v * w.to(dev)).sum().backward()
        xr = x.clone().requires_grad_(True)
        (torch.topk(xr, k).values * w).sum().backward()
        # with a tie straddling the cut torch may pick either index; the kernel takes the lower one: compare as multisets per value
        gd, gr = xd.grad.cpu(), xr.grad
        assert int((gd != 0).sum()) <= k and abs(float(gd.sum() - gr.sum())) < 1e-4
        keep = x != x[n // 2]
        assert torch.equal(gd[keep], gr[keep]), "gradient scatter"
    # hinge generator loss over the k best fakes == reference formula
    x = torch.randn(32, generator=g)
    loss = SL.g_hinge(SL.topk_values(x.to(dev), 10))
    assert abs(float(loss) - float(-torch.mean(torch.topk(x, 10).values))) < 1e-6
    k = 256
    for _ in range(200):
        k = SL.adjust_k(current_k=k, topk_gamma=0.99, inf_k=int(256 * 0.5))
    assert k == 128


def test_prdc_on_device_vs_reference_fixture_and_oracle(sg):
    """metrics.compute_prdc (blocked fp32 GEMM + streaming k-th-value / ball-membership kernels) against (a) the REAL reference's
    compute_prdc output stored in tests/golden/metrics_host.npz (src/metrics/prdc.py:143-168; counts of strict inequalities: exact),
    (b) the float64 oracle restatement at Inception-like scale (3000 x 2048 features, several row blocks, k = 5 and 3)."""
    import os
    from studiogan_amd import metrics as M
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_host.npz"))
    dev = torch.device("cuda:0")
    real, fake = torch.from_numpy(z["in/prdc_real"]), torch.from_numpy(z["in/prdc_fake"])
    for block in (4096, 128):                      # one row block / several (ragged last block)
        m = M.compute_prdc(real.to(dev), fake.to(dev), 5, block=block)
        for k in ("precision", "recall", "density", "coverage"):
            assert abs(m[k] - float(z["exp/prdc_" + k])) < 1e-12, (k, block, m[k], float(z["exp/prdc_" + k]))
    g = torch.Generator().manual_seed(33)
    base = torch.rand(1, 2048, generator=g)
    real = torch.relu(torch.randn(3000, 2048, generator=g) * 0.5 + base)            # non-negative, Inception-pool-like magnitudes
    fake = torch.relu(torch.randn(2500, 2048, generator=g) * 0.55 + base + 0.02)
    for k in (5, 3):
        mo = OI.prdc(real, fake, k)
        m = M.compute_prdc(real.to(dev), fake.to(dev), k, block=1024)
        p, r, d, c = M.calculate_pr_dc(real.to(dev), fake.to(dev), 2500, k)
        assert (p, r, d, c) == (m["precision"], m["recall"], m["density"], m["coverage"])
        for name in ("precision", "recall", "density", "coverage"):
            # a decision flips only when a squared distance ties a radius within fp32 rounding: at most a few of 2500-3000 samples
            assert abs(m[name] - mo[name]) <= 2e-3, (k, name, m[name], mo[name])


def test_frechet_distance_on_device_vs_reference_and_scipy(sg):
    """frechet_inception_distance_device (fp64 Cholesky + B = L2^T L1 + one-sided Jacobi, csrc/linalg.hip) against (a) the REAL
    reference's frechet_inception_distance outputs of tests/golden/metrics_host.npz (src/metrics/fid.py:34-62, scipy sqrtm), including
    the rank-deficient pair that must take the host route, (b) the scipy formula on random 384-dimensional covariances (odd size too),
    (c) the known answer FID(x, x) = 0."""
    import os
    from studiogan_amd import metrics as M
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_host.npz"))
    ab = M.frechet_inception_distance_device(z["in/mu_a"], z["in/sigma_a"], z["in/mu_b"], z["in/sigma_b"])
    assert abs(ab - float(z["exp/fid_ab"])) <= 1e-9 * abs(float(z["exp/fid_ab"])), (ab, float(z["exp/fid_ab"]))
    ac = M.frechet_inception_distance_device(z["in/mu_a"], z["in/sigma_a"], z["in/mu_c"], z["in/sigma_c"])      # sigma_c: rank <= 29 of 64 -> host route
    assert abs(ac - float(z["exp/fid_ac"])) <= 1e-9 * abs(float(z["exp/fid_ac"])), (ac, float(z["exp/fid_ac"]))
    aa = M.frechet_inception_distance_device(z["in/mu_a"], z["in/sigma_a"], z["in/mu_a"], z["in/sigma_a"])
    assert abs(aa) < 1e-9
    rs = np.random.RandomState(3)
    for n in (384, 129):
        xa = rs.randn(4 * n, n) @ (np.eye(n) + 0.2 * rs.randn(n, n))
        xb = rs.randn(4 * n, n) * 1.2 + 0.1
        m1, s1, m2, s2 = xa.mean(0), np.cov(xa, rowvar=False), xb.mean(0), np.cov(xb, rowvar=False)
        ref = float(M.frechet_inception_distance(m1, s1, m2, s2))
        got = M.frechet_inception_distance_device(m1, s1, m2, s2)
        assert abs(got - ref) <= 1e-8 * abs(ref), (n, got, ref)


def test_device_dataset_basket_is_bit_exact(sg):
    """data.DeviceDataset (uint8 data set resident in HBM, reference storage format of utils/hdf5.py / data_util.py): gather + horizontal flip
    bit-identical to numpy indexing, the basket split like torch.split (src/worker.py:194-208), every sample once per epoch, and the
    discriminator consumes a uint8 micro-batch exactly like the host-normalised fp32 batch."""
    from studiogan_amd.data import DeviceDataset
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    imgs = torch.randint(0, 256, (50, 16, 16, 3), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, 10, (50,), generator=g)
    ds = DeviceDataset(imgs.numpy(), labels.numpy(), device=dev, random_flip=True, seed=3)
    idx = torch.tensor([7, 0, 49, 7, 13])
    flip = torch.tensor([1, 0, 1, 0, 0], dtype=torch.uint8)
    out, lab = ds.gather(idx, flip)
    ref = imgs[idx].clone()
    ref[flip.bool()] = ref[flip.bool()].flip(2)
    assert torch.equal(out.cpu(), ref) and torch.equal(lab.cpu(), labels[idx])
    seen = []
    for _ in range(3):                                   # 3 baskets of 2 x 8 = 48 of 50 samples: one epoch, no repeats
        xb, yb = ds.sample_data_basket(8, 2)
        assert len(xb) == 2 and xb[0].shape == (8, 16, 16, 3) and xb[0].dtype == torch.uint8 and yb[1].shape == (8,)
        for x, y in zip(xb, yb):
            for i in range(8):
                xi = x[i].cpu()
                hit = [n for n in range(50) if (torch.equal(imgs[n], xi) or torch.equal(imgs[n].flip(1), xi)) and int(labels[n]) == int(y[i])]
                assert hit, "a basket image must be a (possibly mirrored) data set image with its label"
                seen.append(hit[0])
    assert len(set(seen)) == 48 and ds.epoch == 1
    ds.sample_data_basket(8, 2)
    assert ds.epoch == 2                                  # the next basket does not fit the remaining 2 samples: new permutation


@pytest.mark.parametrize("filt,resizer", [("bicubic", "clean"), ("bilinear", "friendly")])
@pytest.mark.parametrize("src", [32, 128, 512])
def test_pil_resizers_match_pillow(sg, filt, resizer, src):
    """metrics.preprocess_pil against the reference's own resizer functions (src/utils/resize.py:68-78 build_resizer / make_resizer: PIL
    Image.resize on mode 'F' channels) applied to the reference-quantised uint8 image, then (x / 255 - 0.5) / 0.5 as utils/ops.py:258-262:
    up-sampling (32, 128 -> 299) and antialiased reduction (512 -> 299)."""
    from PIL import Image
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(src)
    x = torch.rand(2, 3, src, src, generator=g) * 2.2 - 1.1
    _, q = OI.quantize_resize_normalize(x, quantize=True, size=8)            # q: the reference's uint8 quantisation (bit-exact, pinned elsewhere)
    flt = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}[filt]
    ref = np.zeros((2, 299, 299, 3), dtype=np.float32)
    for n in range(2):
        for c in range(3):
            img = Image.fromarray(q[n, c].astype(np.float32), mode="F").resize((299, 299), resample=flt)
            ref[n, :, :, c] = np.asarray(img)
    ref = (ref / 255.0 - 0.5) / 0.5
    out = M.preprocess_pil(x.to(dev), torch.float32, filt, True, 299)
    torch.cuda.synchronize()
    check(f"PIL {filt} {src}->299", out.cpu(), torch.from_numpy(ref), 2e-6)
    model = M.LoadEvalModel("InceptionV3_tf", resizer, 1, False, dev, state_dict=OI.random_state_dict(0), dtype=torch.bfloat16)
    f, l = model.get_outputs(x.to(dev), quantize=True)
    assert f.shape == (2, 2048) and bool(torch.isfinite(f).all())


@pytest.mark.parametrize("cfg", [(0, 3, 2, 0), (1, 3, 1, 1), (2, 3, 1, 1), (0, 2, 2, 0)])
def test_pool2d_vector_kernel_matches_torch_and_scalar(sg, cfg):
    """sg_pool2d in bf16 (k_pool2d_v8: 8 channels per thread) writing a channel slice of a wider concat tensor, against torch's pooling of
    the same bf16 values (mode 0 max, 1 avg incl. padding, 2 avg excl. padding -- the three poolings of InceptionV3) and against the fp32
    kernel on the up-converted input (same accumulation order: equal after rounding)."""
    import torch.nn.functional as TF
    from studiogan_amd import _lib as L
    mode, k, stride, pad = cfg
    d = torch.device("cuda:0")
    N, H, W, C, ldy, coff = 3, 17, 17, 64, 96, 24
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
    OH = (H + 2 * pad - k) // stride + 1
    xr = x.float().permute(0, 3, 1, 2)
    if mode == 0:
        ref = TF.max_pool2d(xr, k, stride, pad)
    else:
        ref = TF.avg_pool2d(xr, k, stride, pad, count_include_pad=(mode == 1))
    xd = x.to(d)
    y = torch.zeros(N, OH, OH, ldy, dtype=torch.bfloat16, device=d)
    L.call("sg_pool2d", L.BF16, L.ptr(xd), L.ptr(y), N, H, W, C, k, stride, pad, mode, ldy, coff, L.stream())
    y32 = torch.zeros(N, OH, OH, ldy, dtype=torch.float32, device=d)
    x32 = xd.float()
    L.call("sg_pool2d", L.F32, L.ptr(x32), L.ptr(y32), N, H, W, C, k, stride, pad, mode, ldy, coff, L.stream())
    # (round 6) fp32 takes the 4-channels-per-thread kernel k_pool2d_f4; SG_POOL_SCALAR=1 selects the scalar one: bit-identical, and equal to torch's fp32 pooling
    import os
    y32s = torch.zeros_like(y32)
    os.environ["SG_POOL_SCALAR"] = "1"
    try:
        L.call("sg_pool2d", L.F32, L.ptr(x32), L.ptr(y32s), N, H, W, C, k, stride, pad, mode, ldy, coff, L.stream())
    finally:
        del os.environ["SG_POOL_SCALAR"]
    torch.cuda.synchronize()
    assert torch.equal(y32, y32s), "fp32 vector and scalar pooling kernels disagree"
    check(f"pool2d fp32 {cfg}", y32[..., coff:coff + C].cpu().permute(0, 3, 1, 2), ref, 1e-6)
    out = y[..., coff:coff + C].float().cpu().permute(0, 3, 1, 2)
    check(f"pool2d bf16 {cfg}", out, ref.to(torch.bfloat16).float(), 1e-6 if mode == 0 else 4e-3)
    assert torch.equal(y[..., coff:coff + C], y32[..., coff:coff + C].to(torch.bfloat16)), "vector and scalar kernels disagree"
    assert float(y[..., :coff].abs().max()) == 0.0 and float(y[..., coff + C:].abs().max()) == 0.0, "wrote outside its channel slice"


def test_inception_scalar_known_answer(sg):
    """FID InceptionV3 on the HIP path against the convolution-free known answer of tests/inception_kat.py: centre-tap kernels + images that are
    constant per channel collapse the network to a per-channel scalar recursion written from the block wiring alone. Catches what a
    random-weight comparison against a structurally identical oracle could share with it: a swapped branch, a wrong concatenation order,
    a mis-folded batch norm."""
    import inception_kat as K
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    sd = K.centre_tap_state_dict(M.SPEC, 3)
    vals = [[0.3, -0.5, 0.8], [-0.9, 0.1, 0.4]]
    x = torch.tensor(vals).view(2, 3, 1, 1).expand(2, 3, 299, 299)
    model = M.InceptionV3(sd, dev, torch.float32)
    feat, logit = model.forward_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    for i, v in enumerate(vals):
        fk, lk = K.scalar_forward(sd, v)
        check(f"scalar KAT features, image {i}", feat[i], fk, 1e-5)
        check(f"scalar KAT logits, image {i}", logit[i], lk, 1e-5)


def test_load_eval_model_rejects_foreign_weights(sg):
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    sd = M.synthetic_state_dict(0)
    m = M.LoadEvalModel(device=dev, state_dict=sd)
    assert m.weights_pinned is False and m.weights_sha256 is None        # structure checked, provenance unknown
    bad = dict(sd)
    bad.pop("Mixed_7c.branch_pool.conv.weight")
    with pytest.raises(RuntimeError, match="not an FID InceptionV3 state_dict"):
        M.LoadEvalModel(device=dev, state_dict=bad)


def test_frechet_distance_on_device_at_2048_dimensions(sg):
    """the production size: tr sqrtm(S1 S2) of two 2048 x 2048 covariances on the device (fp64 Cholesky + one-sided Jacobi) against scipy's
    sqrtm route of the reference (src/metrics/fid.py:34-62); the device time is written to gpurun_out/ when that directory is writable."""
    import os
    import time
    from studiogan_amd import metrics as M
    rs = np.random.RandomState(7)
    n = 2048
    xa = rs.randn(3 * n, n) @ (np.eye(n) + 0.05 * rs.randn(n, n))
    xb = rs.randn(3 * n, n) * 1.1 + 0.05
    m1, s1, m2, s2 = xa.mean(0), np.cov(xa, rowvar=False), xb.mean(0), np.cov(xb, rowvar=False)
    t0 = time.time()
    ref = float(M.frechet_inception_distance(m1, s1, m2, s2))
    t_host = time.time() - t0
    M.frechet_inception_distance_device(m1[:64], s1[:64, :64], m2[:64], s2[:64, :64])      # warm-up (module load)
    torch.cuda.synchronize()
    t0 = time.time()
    got = M.frechet_inception_distance_device(m1, s1, m2, s2)
    torch.cuda.synchronize()
    t_dev = time.time() - t0
    line = f"FID back-end at n = 2048: device {t_dev:.2f} s (fp64 Cholesky + Jacobi), host scipy.linalg.sqrtm {t_host:.2f} s; values {got:.9f} vs {ref:.9f}"
    print(line)
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "fid_backend_2048.txt"), "w").write(line + "\n")
    except OSError:
        pass
    assert abs(got - ref) <= 1e-7 * abs(ref), (got, ref)
