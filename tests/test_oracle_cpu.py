"""CPU: the oracle restatement against the committed golden vectors (produced by the real reference), against the
reference itself when /root/reference is present, and analytic known-answer tests."""
import copy
import os
import sys

import pytest
import torch

from util import check, load_golden, sub
from oracle import restate as O
from oracle import make_golden as MG


def _run_restatement(name):
    fix, meta = load_golden(name)
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    GI, DI = sub(fix, "G_init/"), sub(fix, "D_init/")
    pnames = lambda d: {k: v.clone() for k, v in d.items() if not any(s in k for s in ("weight_u", "weight_v", "running_", "num_batches"))}
    bnames = lambda d: {k: v.clone() for k, v in d.items() if any(s in k for s in ("weight_u", "weight_v", "running_", "num_batches"))}
    GP, GB, DP, DB = pnames(GI), bnames(GI), pnames(DI), bnames(DI)
    ins = sub(fix, "in/")
    exp = MG.run_restatement(ocfg, y, GP, GB, DP, DB, ins, meta["n_d"], meta["seed"])
    return fix, exp


@pytest.mark.parametrize("name", ["biggan32", "sngan32", "resgan32", "dcgan32", "sndcgan32", "wgangp32", "sngp32", "bigdeep32", "bigdeepsg32",
                                  "biggan128w", "sngan32w"])      # (wgangp128w / bigdeep128w: bit-identity asserted when oracle/make_golden.py wrote them)
def test_restatement_matches_golden(name):
    fix, exp = _run_restatement(name)
    gold = sub(fix, "exp/")
    assert set(gold.keys()) == set(exp.keys())
    for k in sorted(gold.keys()):
        check(k, exp[k], gold[k], 1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["biggan32", "sngan32", "dcgan32", "wgangp32", "bigdeep32", "bigdeepsg32", "biggan128w"])
def test_golden_regenerates_from_reference(name):
    """The committed fixture is exactly what the reference produces today (guards against stale fixtures)."""
    from oracle import ref_import as R
    fix, meta = load_golden(name)
    c = MG.CONFIGS[name]
    cfgs = R.load_cfgs(c["yaml"])
    torch.manual_seed(c["seed"])
    Gen, Dis = R.build_models(cfgs)
    if c.get("compact"):
        Gen.load_state_dict(MG.formula_state(meta["G_spec"], c["seed"]), strict=True)
        Dis.load_state_dict(MG.formula_state(meta["D_spec"], c["seed"] + 1), strict=True)
    ocfg = MG.oracle_cfg(c["yaml"])
    ins = MG.synth_inputs(c["seed"] + 1, c["n_d"], c["batch"], ocfg["z_dim"], ocfg["num_classes"], ocfg["img_size"])
    exp = MG.run_reference(cfgs, Gen, Dis, ins, c["n_d"], c["seed"])
    for k in ("fake0", "adv_r0", "d_loss0", "g_loss"):
        check("ref:" + k, exp[k], fix["exp/" + k], 1e-6)


def test_param_counts_match_reference_logs():
    """Known-answer: BigGAN-128 G=70,433,988 / D=87,982,370 (reference logs/IMAGENET/BigGAN256-train-2021_01_24_03_52_15.log:11,122)."""
    from studiogan_amd import ops
    from studiogan_amd.backbones import big_resnet

    class M:
        info_type = "N/A"
    MOD = ops.Modules(apply_g_sn=True, apply_d_sn=True, g_cond_mtd="cBN", backbone="big_resnet")
    G = big_resnet.Generator(120, 128, 128, 96, True, [4], "cBN", 1000, "ortho", "N/A", True, MOD, M)
    D = big_resnet.Discriminator(128, 96, True, True, [1], "PD", "W/O", "N/A", False, 1000, "ortho", "N/A", True, MOD, M)
    assert sum(p.numel() for p in G.parameters()) == 70433988
    assert sum(p.numel() for p in D.parameters()) == 87982370


def test_analytic_kats():
    z = torch.zeros(8)
    assert float(O.d_loss("hinge", z, z)) == 2.0            # utils/losses.py:226-227
    assert float(O.g_loss("hinge", z)) == 0.0               # utils/losses.py:230-231
    assert abs(float(O.d_loss("vanilla", z, z)) - 2 * 0.6931471805599453) < 1e-6
    # SelfAttention is the identity at initialisation (sigma = 0, utils/ops.py:81,103)
    torch.manual_seed(0)
    P = {"a.conv1x1_theta.weight": torch.randn(2, 16, 1, 1), "a.conv1x1_phi.weight": torch.randn(2, 16, 1, 1),
         "a.conv1x1_g.weight": torch.randn(8, 16, 1, 1), "a.conv1x1_attn.weight": torch.randn(16, 8, 1, 1), "a.sigma": torch.zeros(1)}
    x = torch.randn(2, 16, 8, 8)
    assert torch.equal(O.self_attention(x, P, {}, "a"), x)
    # spectral norm: sigma of the normalised weight is ~1 after convergence
    P = {"l.weight_orig": torch.randn(12, 20)}
    B = {"l.weight_u": torch.nn.functional.normalize(torch.randn(12), dim=0), "l.weight_v": torch.nn.functional.normalize(torch.randn(20), dim=0)}
    for _ in range(200):
        w = O.weight_of(P, B, "l")
    assert abs(float(torch.linalg.matrix_norm(w, 2)) - 1.0) < 1e-4


def test_inception_oracle_structure_and_kats():
    from oracle import inception as OI
    sd = OI.random_state_dict(0)
    assert len([k for k in sd if k.endswith(".conv.weight")]) == 94          # torchvision inception_v3 has 94 BasicConv2d
    assert sd["fc.weight"].shape == (1008, 2048)                              # FID weights: 1008-way fc (inception_net.py:117)
    x = torch.rand(1, 3, 299, 299) * 2 - 1
    feat, logits = OI.inception_forward(x, sd)
    assert feat.shape == (1, 2048) and logits.shape == (1, 1008) and bool((feat >= 0).all())
    u = torch.full((32, 1008), 1.0 / 1008)
    assert abs(float(OI.inception_score(u)[0]) - 1.0) < 1e-6                  # IS of uniform predictions = 1 (ins.py:28-42)
    import numpy as np
    mu, s = np.arange(4.0), np.eye(4) * 2
    assert abs(OI.frechet_distance(mu, s, mu, s)) < 1e-9                      # FID(x,x) = 0 (fid.py:34-62)
    imgs = torch.tensor([[[[-1.0, 1.0], [0.0, 0.00392]]]]).repeat(1, 3, 1, 1)
    _, q = OI.quantize_resize_normalize(imgs, size=4)
    assert q[0, 0].tolist() == [[0, 255], [128, 128]]                         # uint8 truncation of (x+1)/2*255+0.5


def test_regulariser_restatements_match_reference_fixture():
    """tests/golden/regularisers.npz holds the outputs of the REAL reference functions (utils/losses.py cal_maxgrad_penalty,
    cal_dra_penalty, cal_r1_reg, lecam_reg, adjust_k; utils/ops.py LeCamEMA; the ToTensor + Normalize input transform) on a seeded SN +
    projection discriminator -- produced by oracle/make_golden_regularisers.py, which also asserts bit-identity at generation time.
    Here the restatements (the checker of the GPU tests) are re-run on the stored inputs."""
    fix, meta = load_golden("regularisers")
    ocfg = MG.oracle_cfg(meta["yaml"])
    dis_fn = O.model_fns(ocfg)[1]
    DP, DB = sub(fix, "D_P/"), sub(fix, "D_B/")
    real, fake, lab = fix["in/real"], fix["in/fake"], fix["in/lab"]

    def run(name, fn):
        leaves = {k: v.clone().requires_grad_(True) for k, v in DP.items()}
        out = fn(leaves, {k: v.clone() for k, v in DB.items()})
        gs = torch.autograd.grad(out, list(leaves.values()), allow_unused=True)
        check(name, out.detach(), fix["exp/" + name], 1e-6)
        gmax = max(float(fix[f"exp/{name}_grad/" + k].abs().max()) for k in leaves)
        for k, g in zip(leaves, gs):
            ref = fix[f"exp/{name}_grad/" + k]
            g = torch.zeros_like(ref) if g is None else g
            assert float((g - ref).abs().max()) <= 1e-5 * gmax, f"{name} grad {k}"

    run("maxgp", lambda P, B: O.maxgrad_penalty(dis_fn, real, lab, fake, P, B, fix["in/maxgp_alpha"]))
    run("dra", lambda P, B: O.dra_penalty(dis_fn, real, lab, P, B, fix["in/dra_alpha"], fix["in/dra_noise"]))
    run("r1", lambda P, B: O.r1_reg(dis_fn, real, lab, P, B)[0])
    # LeCam with the EMA values the reference's LeCamEMA reached after the recorded update sequence; the product-side EMA class too
    from studiogan_amd import ops, losses as SL
    ema = ops.LeCamEMA(decay=0.9, start_iter=2)
    for cur, mode, itr in meta["lecam_updates"]:
        ema.update(cur, mode, itr)
    assert [ema.D_real, ema.D_fake] == [float(v) for v in fix["exp/lecam_ema"]]
    a, b = fix["in/lecam_real"].clone().requires_grad_(True), fix["in/lecam_fake"].clone().requires_grad_(True)
    lo = O.lecam_reg(a, b, ema.D_real, ema.D_fake)
    lo.backward()
    assert torch.equal(lo.detach(), fix["exp/lecam"]) and torch.equal(a.grad, fix["exp/lecam_dreal"]) and torch.equal(b.grad, fix["exp/lecam_dfake"])
    # top-k schedule and the uint8 input transform
    k, ks = 64, []
    for _ in range(100):
        k = SL.adjust_k(current_k=k, topk_gamma=0.99, inf_k=int(64 * 0.5))
        ks.append(k)
    assert ks == [float(v) for v in fix["exp/adjust_k"]]
    assert torch.equal(torch.topk(fix["in/topk_x"], 10).values, fix["exp/topk_10"])
    assert torch.equal(O.uint8_to_normalized(fix["in/u8"]), fix["exp/u8_norm"])


def test_metric_host_formulas_match_reference_fixture():
    """FID (reference src/metrics/fid.py:34-62, incl. the near-singular eps branch) and the Inception score (src/metrics/ins.py:28-42):
    the product's host-side functions and the oracle's FID against outputs of the REAL reference functions
    (oracle/make_golden_metrics.py -> tests/golden/metrics_host.npz)."""
    import numpy as np
    from studiogan_amd import metrics as M
    from oracle import inception as OI
    fix, _ = load_golden("metrics_host")
    g = lambda k: fix[k].numpy()
    for pair in ("ab", "ac", "aa"):
        x, y = pair
        ref = float(fix["exp/fid_" + pair])
        ours = float(M.frechet_inception_distance(g("in/mu_" + x), g("in/sigma_" + x), g("in/mu_" + y), g("in/sigma_" + y)))
        orc = float(OI.frechet_distance(g("in/mu_" + x), g("in/sigma_" + x), g("in/mu_" + y), g("in/sigma_" + y)))
        assert abs(ours - ref) <= 1e-9 * max(1.0, abs(ref)), (pair, ours, ref)
        assert abs(orc - ref) <= 1e-6 * max(1.0, abs(ref)), (pair, orc, ref)
    p = fix["in/probs"]
    for splits in (1, 5):
        m, s = M.calculate_kl_div(p, splits)
        assert abs(float(m) - float(fix[f"exp/is_mean_{splits}"])) < 1e-6 and (splits == 1 or abs(float(s) - float(fix[f"exp/is_std_{splits}"])) < 1e-6)
    # evaluation pre-processing: uint8 quantisation bit-exact, legacy bilinear resize + normalisation sampled + norms
    import numpy as np
    x = fix["in/pre_x"]
    r, q = OI.quantize_resize_normalize(x, quantize=True)
    assert np.array_equal(q, fix["exp/pre_q"].numpy())
    flat = r.reshape(-1)
    idx = (torch.arange(4096, dtype=torch.int64) * flat.numel()) // 4096
    assert torch.equal(flat[idx], fix["exp/pre_samples"])
    n = fix["exp/pre_norms"]
    assert abs(float(flat.double().sum()) - float(n[0])) <= 1e-9 * abs(float(n[0])) and abs(float(flat.double().norm()) - float(n[1])) <= 1e-12 * float(n[1])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_is_accuracy_label_remap_matches_reference():
    """IS top-1/top-5 label handling (reference src/metrics/ins.py:45-79 + src/utils/misc.py:582-595, TF-Inception on ImageNet): the
    product's host logic (label table, loader-label -> folder -> TF index, the [:, 1:1001] slice) against the reference's own
    eval_features on the same synthetic probabilities (the device top-k kernel is replaced by a CPU stand-in with sklearn's tie rule,
    which tests/test_eval_gpu.py checks bit-exactly against the kernel)."""
    import importlib
    import types
    import numpy as np
    from oracle import ref_import as RI
    from studiogan_amd import metrics as M
    RI._prepare()
    ins = importlib.import_module("metrics.ins")
    table = "/root/reference/src/utils/tf_imagenet_folder_label_pairs.txt"
    d = M.load_ImageNet_label_dict(table)
    assert len(d) == 1000 and d["n02119789"] == 0 and d["n02100735"] == 1
    folders = sorted(d.keys())                                   # ImageFolder sorts class folders alphabetically
    class_to_idx = {f: i for i, f in enumerate(folders)}
    g = torch.Generator().manual_seed(12)
    n = 3000
    labels = torch.cat([torch.arange(1000), torch.randint(0, 1000, (n - 1000,), generator=g)]).tolist()   # every class present (sklearn needs it)
    logits = torch.randn(n, 1008, generator=g)
    for i, l in enumerate(labels):                               # make ~half of the samples right
        if i % 2 == 0:
            logits[i, 1 + d[folders[l]]] += 4.0
    probs = torch.softmax(logits, 1)
    loader = types.SimpleNamespace(dataset=types.SimpleNamespace(data_name="ImageNet", data=types.SimpleNamespace(class_to_idx=class_to_idx)))
    cwd = os.getcwd()
    os.chdir("/root/reference")
    try:
        ref = ins.eval_features(probs, labels, loader, n, 1, True, is_torch_backbone=False)
    finally:
        os.chdir(cwd)

    def cpu_topk(p, lab, k, c0, ncls):      # sklearn's rule: stable ascending argsort reversed -> the higher class index wins a tie
        s = p[:, c0:c0 + ncls].numpy()
        t = np.asarray(lab)
        st = s[np.arange(len(t)), t][:, None]
        beat = (s > st).sum(1) + ((s == st) & (np.arange(ncls)[None, :] > t[:, None])).sum(1)
        return float((beat < k).mean())
    ours = M.eval_features(probs, labels, n, 1, True, class_to_idx=class_to_idx, folder_label_dict=d, topk_fn=cpu_topk)
    assert abs(float(ours[0]) - float(ref[0])) < 1e-6
    assert abs(ours[2] - ref[2]) < 1e-12 and abs(ours[3] - ref[3]) < 1e-12, (ours, ref)
    assert 0.2 < ours[2] < 0.7 and ours[3] >= ours[2]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_ema_restatement_matches_reference_class():
    """oracle.ema_update against the reference's Ema (src/utils/ema.py:11-40) on a small network with batch-norm buffers: before
    start_iter (decay 0 -> copy), after it (lerp), and the integer buffer (copied, never lerped)."""
    import importlib
    import torch.nn as nn
    from oracle import ref_import as RI
    RI._prepare()
    ref_ema = importlib.import_module("utils.ema")

    def net(seed):
        torch.manual_seed(seed)
        m = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Conv2d(4, 2, 1))
        m.train()
        m(torch.randn(5, 3, 6, 6))          # moves running stats and num_batches_tracked
        return m
    src, tgt = net(1), net(2)
    e = ref_ema.Ema(src, tgt, decay=0.9, start_iter=3)       # copies src into tgt
    oP = {k: v.detach().clone() for k, v in tgt.named_parameters()}
    oB = {k: v.detach().clone() for k, v in tgt.named_buffers()}
    for it in (0, 2, 3, 7):
        with torch.no_grad():                                # the source trains on
            for p in src.parameters():
                p.add_(0.1 * torch.randn_like(p))
        src(torch.randn(5, 3, 6, 6))
        e.update(it)
        O.ema_update(dict(src.named_parameters()), dict(src.named_buffers()), oP, oB, it, decay=0.9, start_iter=3)
        for k, v in tgt.named_parameters():
            assert torch.equal(v, oP[k]), (it, k)
        for k, v in tgt.named_buffers():
            assert torch.equal(v, oB[k]), (it, k)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_inception_oracle_pinned_to_the_references_own_code():
    """oracle/inception.py against the REFERENCE's code (src/metrics/inception_net.py): fid_inception_v3(), the FID-patched
    FIDInceptionA / C / E_1 / E_2.forward (:135-249) and InceptionV3.__init__ / forward (:16-107) run unmodified on CPU with
    oracle/tv_stub.py standing in for the absent torchvision (its published BasicConv2d / Inception A-E constructors) and the
    oracle's seeded weights returned by the (patched) checkpoint download. Bit-identical block by block and end to end. What stays
    unpinned: the values of the real FID checkpoint (not obtainable offline)."""
    import importlib
    from oracle import ref_import as R, tv_stub, inception as OI
    R._prepare()
    names = ("torchvision", "torchvision.models", "torchvision.models.inception", "torchvision.models.utils", "metrics.inception_net")
    saved = {k: sys.modules.get(k) for k in names}
    try:
        tv, models, inc = tv_stub.as_modules()
        sys.modules["torchvision"], sys.modules["torchvision.models"], sys.modules["torchvision.models.inception"] = tv, models, inc
        sys.modules.pop("torchvision.models.utils", None)
        sys.modules.pop("metrics.inception_net", None)
        net = importlib.import_module("metrics.inception_net")
        sd = OI.random_state_dict(5)
        net.load_state_dict_from_url = lambda *a, **k: sd        # inception_net.py:130 downloads pt_inception-2015-12-05-6726825d.pth
        ref = net.InceptionV3(resize_input=False, normalize_input=False).eval()      # metrics/preparation.py:53
        g = torch.Generator().manual_seed(2)
        with torch.no_grad():
            # block level: the reference's patched forwards on their own inputs
            _, inception = net.fid_inception_v3()
            inception.eval()
            for name, cin, hw, fn in (("Mixed_5b", 192, 12, lambda x, p: OI._blockA(x, sd, p)), ("Mixed_5d", 288, 12, lambda x, p: OI._blockA(x, sd, p)),
                                      ("Mixed_6a", 288, 13, lambda x, p: OI._blockB(x, sd, p)), ("Mixed_6c", 768, 9, lambda x, p: OI._blockC(x, sd, p)),
                                      ("Mixed_7a", 768, 9, lambda x, p: OI._blockD(x, sd, p)), ("Mixed_7b", 1280, 5, lambda x, p: OI._blockE(x, sd, p, "avg")),
                                      ("Mixed_7c", 2048, 5, lambda x, p: OI._blockE(x, sd, p, "max"))):
                x = torch.randn(2, cin, hw, hw, generator=g)
                a, b = getattr(inception, name)(x), fn(x, name)
                assert a.shape == b.shape and torch.equal(a, b), name
            assert type(inception.Mixed_7b).__name__ == "FIDInceptionE_1" and type(inception.Mixed_7c).__name__ == "FIDInceptionE_2"
            # end to end, at the real input size
            x = torch.rand(2, 3, 299, 299, generator=g) * 2 - 1
            f_ref, l_ref = ref(x)
            f_o, l_o = OI.inception_forward(x, sd)
        assert f_ref.shape == (2, 2048) and l_ref.shape == (2, 1008)
        assert torch.equal(f_ref, f_o) and torch.equal(l_ref, l_o)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_prdc_restatement_matches_reference_fixture():
    """oracle/inception.py prdc() on the stored inputs against the REAL reference's compute_prdc outputs (tests/golden/metrics_host.npz,
    written by oracle/make_golden_metrics.py, which also asserts the identity at generation time)."""
    import numpy as np
    from oracle import inception as OI
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_host.npz"))
    m = OI.prdc(torch.from_numpy(z["in/prdc_real"]), torch.from_numpy(z["in/prdc_fake"]), 5)
    for k in ("precision", "recall", "density", "coverage"):
        assert abs(m[k] - float(z["exp/prdc_" + k])) < 1e-12, (k, m[k])


def test_heads_restatement_matches_reference_fixture():
    """oracle/restate.py d_heads / cond_loss / crammer_singer / d_side_loss on the stored states and inputs against the REAL reference's
    outputs (tests/golden/heads.npz: big_resnet.Discriminator heads + utils/losses.py, combined as src/worker.py:281-317)."""
    import json
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    meta = json.load(open(os.path.join(here, "golden", "heads.json")))
    z = np.load(os.path.join(here, "golden", "heads.npz"))
    for name, c in meta["cases"].items():
        y = c["yaml"]
        ocfg = dict(MG.oracle_cfg(y), num_classes=y["DATA"]["num_classes"])
        P = {k[len(name) + 3:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith(name + "/P/")}
        B = {k[len(name) + 3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(name + "/B/")}
        get = lambda k: torch.from_numpy(z[name + "/in/" + k])
        loss, rd, fd = O.d_side_loss(O.model_fns(ocfg)[1], P, B, ocfg, get("real"), get("rl"), get("fake"), get("fl"), c["adv_loss"], meta["hp"])
        loss.backward()
        assert abs(float(loss.detach()) - float(z[name + "/exp/loss"])) < 1e-5, name
        for k, p in P.items():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            assert float((g - torch.from_numpy(z[name + "/grad/" + k])).abs().max()) < 1e-5, (name, k)


def test_inception_oracle_matches_the_convolution_free_known_answer():
    """oracle/inception.py against tests/inception_kat.py: with centre-tap kernels and an image that is constant per channel the FID network is a
    per-channel scalar recursion (no convolution or pooling implementation involved) -- an independent statement of the block wiring
    (reference src/metrics/inception_net.py:117-127,135-249 over torchvision's inception.py) that the oracle and, through
    tests/test_eval_gpu.py::test_inception_scalar_known_answer, the HIP path must both reproduce."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import inception_kat as K
    from oracle import inception as OI
    sd = K.centre_tap_state_dict(OI.SPEC, 3)
    vals = [0.3, -0.5, 0.8]
    x = torch.tensor(vals).view(1, 3, 1, 1).expand(1, 3, 299, 299).contiguous()
    f, lg = OI.inception_forward(x, sd)
    fk, lk = K.scalar_forward(sd, vals)
    assert int((fk > 0).sum()) > 500, "the known answer must exercise the network (most features non-zero)"
    assert float((f[0].double() - fk).abs().max()) <= 1e-5 * float(fk.abs().max())
    assert float((lg[0].double() - lk).abs().max()) <= 1e-5 * float(lk.abs().max())


def test_inception_manifest_and_weight_file_pin(tmp_path):
    """studiogan_amd.metrics: the strict structural check of an FID InceptionV3 state_dict and the sha256 check of the published file name
    (reference src/metrics/inception_net.py:13,117-130: load_state_dict_from_url of pt_inception-2015-12-05-6726825d.pth + strict load)."""
    import studiogan_amd  # noqa: F401
    from studiogan_amd import metrics as M
    from oracle import inception as OI
    man = M.inception_manifest()
    sd = OI.random_state_dict(0)
    assert {k for k in sd if not k.endswith("num_batches_tracked")} == set(man)        # the oracle's generator and the manifest agree on the key set
    assert all(tuple(sd[k].shape) == man[k] for k in man)
    assert len(man) == 94 * 5 + 2 and man["fc.weight"] == (1008, 2048)
    assert M.validate_inception_state_dict(sd)
    assert M.validate_inception_state_dict({k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")})   # a TF-converted file has none
    for mutate in (lambda d: d.pop("fc.bias"), lambda d: d.update({"AuxLogits.fc.weight": torch.zeros(1)}),
                   lambda d: d.update({"Conv2d_1a_3x3.conv.weight": torch.zeros(32, 3, 5, 5)}),
                   lambda d: d.update({"fc.weight": torch.zeros(1008, 2048, dtype=torch.int32)})):
        bad = dict(sd)
        mutate(bad)
        with pytest.raises(RuntimeError, match="not an FID InceptionV3 state_dict"):
            M.validate_inception_state_dict(bad)
    # a file that is not the published one: right structure, wrong hash
    path = tmp_path / M.FID_WEIGHTS_FILE
    torch.save(sd, path)
    with pytest.raises(RuntimeError, match="does not start with 6726825d"):
        M.load_fid_weights(str(path))
    sd2, digest = M.load_fid_weights(str(path), check_hash=False)
    assert len(digest) == 64 and set(sd2) == set(sd)
    assert M.FID_WEIGHTS_URL.endswith(M.FID_WEIGHTS_FILE) and M.FID_WEIGHTS_SHA256_PREFIX in M.FID_WEIGHTS_FILE
