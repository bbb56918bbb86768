"""Checks of the augmentation / l2-loss host mirrors (studiogan_amd.diffaug, .cr, losses.l2_loss) against the vectors the reference wrote
(tests/golden/aug.npz, oracle/make_golden_aug.py), written once for both places they run: the GPU (tests/test_aug_gpu.py) and the kernel sources on
the CPU interpreter (tests/test_aug_cpu.py under hipemu.fullemu.Installed). fp32 throughout; tolerance 2e-6 of the expected tensor's range where the
contrast mean's summation order enters, exact equality where it does not."""
import os

import numpy as np
import torch

from util import check
from oracle import aug_ref as AR
from oracle import make_golden_aug as MGA

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aug.npz")
TOL = 2e-6


def _t(z, k):
    return torch.from_numpy(z[k]) if k in z.files else None


def _replay(draws):
    """torch.rand / torch.randint / FloatTensor.uniform_ stand-ins that hand out the recorded draws in order (shape-checked): the host mirrors then
    consume exactly what the reference consumed when it wrote the fixture, whatever generator the device has"""
    it = draws if hasattr(draws, "__next__") else iter(draws)

    def nxt(shape, dev):
        d = next(it)
        assert tuple(d.shape) == tuple(shape), (tuple(d.shape), tuple(shape))
        return d.to(dev)
    return nxt


class Replayed:
    def __init__(self, draws):
        self.it = iter(draws)
        self.nxt = _replay(self.it)

    def __enter__(self):
        self.saved = (torch.rand, torch.randint, torch.FloatTensor, torch.randn)
        nxt = self.nxt

        def rand(*size, dtype=None, device=None, **kw):
            if len(size) == 1 and isinstance(size[0], (list, tuple)):
                size = tuple(size[0])
            return nxt(size, device)

        def randint(low, high, size=None, device=None, **kw):
            return nxt(size, device)

        class FT:
            def __init__(self, *size):
                self.size = size

            def uniform_(self, a=0.0, b=1.0):
                return nxt(self.size, "cpu")
        torch.rand, torch.randint, torch.FloatTensor, torch.randn = rand, randint, FT, rand
        return self

    def __exit__(self, *a):
        torch.rand, torch.randint, torch.FloatTensor, torch.randn = self.saved


def diffaug_case(case, dev):
    """output, gradient and the second-order (linear) term of apply_diffaug against the reference's vectors"""
    from studiogan_amd import diffaug as DA
    tag, shape, policy = case
    z = np.load(GOLD)
    p = f"diffaug/{tag}/"
    draws = []
    while p + f"draw{len(draws)}" in z.files:
        draws.append(_t(z, p + f"draw{len(draws)}"))
    x = _t(z, p + "x").to(dev).requires_grad_(True)
    with Replayed(draws):
        y = DA.apply_diffaug(x, policy)
    exact = "color" not in policy          # no contrast mean: same fp32 operations in the same order as the reference
    if exact:
        assert torch.equal(y.detach().cpu(), _t(z, p + "y")), tag
    check(f"diffaug {tag} y", y, _t(z, p + "y"), TOL)
    gy = _t(z, p + "gy").to(dev).requires_grad_(True)
    (dx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    check(f"diffaug {tag} dx", dx, _t(z, p + "dx"), TOL)
    (lin,) = torch.autograd.grad(dx, gy, _t(z, p + "gg").to(dev))
    check(f"diffaug {tag} second order", lin, _t(z, p + "lin"), TOL)
    # channels-last entry (diffaug.py:37-44)
    with Replayed(draws):
        y2 = DA.apply_diffaug(x.detach().permute(0, 2, 3, 1).contiguous(), policy, channels_first=False)
    assert y2.shape == (shape[0], shape[2], shape[3], shape[1]) and y2.is_contiguous()
    assert torch.equal(y2.permute(0, 3, 1, 2), y.detach())


def cr_case(case, dev):
    from studiogan_amd import cr as CR
    tag, shape, flip, trans = case
    z = np.load(GOLD)
    p = f"cr/{tag}/"
    draws = ([_t(z, p + "coin")] if flip else []) + ([_t(z, p + "tx"), _t(z, p + "ty")] if trans else [])
    x = _t(z, p + "x").to(dev).requires_grad_(True)
    with Replayed(draws):
        y = CR.apply_cr_aug(x, flip=flip, translation=trans)
    assert torch.equal(y.detach().cpu(), _t(z, p + "y")), tag            # a pure gather: bit for bit
    (dx,) = torch.autograd.grad(y, x, _t(z, p + "gy").to(dev))
    check(f"cr {tag} dx", dx, _t(z, p + "dx"), TOL)


def mse_case(case, dev):
    from studiogan_amd import losses
    tag, shape = case
    z = np.load(GOLD)
    p = f"mse/{tag}/"
    a, b = _t(z, p + "a").to(dev).requires_grad_(True), _t(z, p + "b").to(dev).requires_grad_(True)
    loss = losses.l2_loss(a, b)
    assert loss.dim() == 0
    assert abs(float(loss.detach()) - float(z[p + "loss"])) <= 2e-6 * abs(float(z[p + "loss"])), tag
    da, db = torch.autograd.grad(loss, [a, b], torch.tensor(0.7, device=dev))
    check(f"mse {tag} da", da, _t(z, p + "da"), TOL)
    check(f"mse {tag} db", db, _t(z, p + "db"), TOL)
    (da_only,) = torch.autograd.grad(losses.l2_loss(a, b.detach()), a, torch.tensor(0.7, device=dev))     # one-sided (detached partner, worker.py:603)
    assert torch.equal(da_only, da)


def adjoint_and_linearity(shape, ops_policy, dev, seed=0):
    """size-independent properties (run at the benchmark's image sizes on the GPU): <A u, v> == <u, A^T v> for the linear part A (ties the backward kernel
    to the forward one), A(a u1 + b u2) == a A u1 + b A u2, and y(x) - A x == the brightness offset pushed through the rest of the chain"""
    from studiogan_amd import functional as F
    from studiogan_amd import _lib as L
    g = torch.Generator().manual_seed(seed)
    N, C, H, W = shape
    u, u2, v = (torch.randn(shape, generator=g).to(dev) for _ in range(3))
    color = torch.stack([torch.rand(N, generator=g) - 0.5, torch.rand(N, generator=g) * 2, torch.rand(N, generator=g) + 0.5], 1).to(dev)
    mt = max(H // 8, 1)
    geom = torch.stack([torch.randint(-mt, mt + 1, (N,), generator=g), torch.randint(-mt, mt + 1, (N,), generator=g), torch.randint(0, H, (N,), generator=g),
                        torch.randint(0, W, (N,), generator=g), torch.randint(0, 2, (N,), generator=g)], 1).to(torch.int32).to(dev)
    spec = F.AugSpec(ops_policy, color, geom, (H + 1) // 2, (W + 1) // 2, mt)
    Au = F.AugmentFn.apply(u, spec, True)
    Atv = F.AugmentBwdFn.apply(v, spec)
    lhs, rhs = float((Au.double() * v.double()).sum()), float((u.double() * Atv.double()).sum())
    scale = float(Au.double().norm() * v.double().norm()) + 1e-30
    assert abs(lhs - rhs) <= 1e-6 * scale, (lhs, rhs, scale)
    comb = F.AugmentFn.apply(0.75 * u - 1.5 * u2, spec, True)
    check("augment linearity", comb, 0.75 * Au - 1.5 * F.AugmentFn.apply(u2, spec, True), 5e-6)
    if ops_policy & L.AUG_BRIGHTNESS:
        off = F.AugmentFn.apply(u, spec) - Au                     # the offset b pushed through saturation / contrast (identity on a constant image), then the geometry
        ones = F.AugmentFn.apply(torch.ones_like(u), F.AugSpec(ops_policy & ~(L.AUG_BRIGHTNESS | L.AUG_SATURATION | L.AUG_CONTRAST), None, geom, spec.cut_h, spec.cut_w, mt), True) \
            if ops_policy & ~(L.AUG_BRIGHTNESS | L.AUG_SATURATION | L.AUG_CONTRAST) else torch.ones_like(u)
        check("augment brightness offset", off, ones * color[:, 0].reshape(N, 1, 1, 1), 5e-6)
    return Au


def oracle_spec_case(shape, ops, dev, seed=0):
    """sg_augment with explicit tables against the restatement's operators composed in the kernel's fixed order (any operator subset, both translation kinds)"""
    from studiogan_amd import functional as F
    from studiogan_amd import _lib as L
    g = torch.Generator().manual_seed(seed)
    N, C, H, W = shape
    x = (torch.rand(shape, generator=g) * 2 - 1)
    gy = torch.randn(shape, generator=g)
    b, s, c = torch.rand(N, generator=g) - 0.5, torch.rand(N, generator=g) * 2, torch.rand(N, generator=g) + 0.5
    mt = max(min(H, W) // 8, 1)
    tx, ty = torch.randint(-mt, mt + 1, (N,), generator=g), torch.randint(-mt, mt + 1, (N,), generator=g)
    ch, cw = (H + 1) // 2, (W + 1) // 2
    cx, cy = torch.randint(0, H + (1 - ch % 2), (N,), generator=g), torch.randint(0, W + (1 - cw % 2), (N,), generator=g)
    fl = torch.randint(0, 2, (N,), generator=g)
    xr = x.clone().requires_grad_(True)
    e = xr
    if ops & L.AUG_BRIGHTNESS:
        e = AR.brightness(e, b)
    if ops & L.AUG_SATURATION:
        e = AR.saturation(e, s)
    if ops & L.AUG_CONTRAST:
        e = AR.contrast(e, c)
    if ops & L.AUG_FLIP:
        e = AR.cr_flip(e, fl)
    if ops & L.AUG_TRANSLATE:
        e = AR.translation(e, tx, ty)
    if ops & L.AUG_TRANSLATE_REFLECT:
        e = AR.cr_translation(e, tx, ty)
    if ops & L.AUG_CUTOUT:
        e = AR.cutout(e, cx, cy, ch, cw)
    (edx,) = torch.autograd.grad(e, xr, gy)
    color = torch.stack([b, s, c], 1).to(dev)
    geom = torch.stack([tx, ty, cx, cy, fl], 1).to(torch.int32).to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = F.AugmentFn.apply(xd, F.AugSpec(ops, color, geom, ch, cw, mt))
    check(f"augment ops={ops} {shape} y", y, e, TOL)
    (dx,) = torch.autograd.grad(y, xd, gy.to(dev))
    check(f"augment ops={ops} {shape} dx", dx, edx, TOL)


DIFFAUG_CASES, CR_CASES, MSE_CASES = MGA.DIFFAUG_CASES, MGA.CR_CASES, MGA.MSE_CASES


# ---- the worker's discriminator / generator update with DiffAugment + CR / bCR / zCR against the reference's vectors -------------------------------------
def consistency_case(tag, dev):
    """studiogan_amd.worker.Worker with apply_diffaug / apply_cr / apply_bcr / apply_zcr on the networks and inputs of tests/golden/<config>.npz, fed the
    recorded draws: loss value and every parameter gradient of one discriminator update and one generator update against what the REAL reference's
    models + utils/diffaug.py + utils/cr.py + MSELoss produced when combined as src/worker.py:236-365,520-603 (tests/golden/consistency.npz,
    oracle/make_golden_consistency.py). fp32; tolerances of tests/test_model_gpu.py::step_vs_golden."""
    import json
    from util import load_golden, sub, hyper, Collector, GOLDEN
    from test_model_gpu import build_from_yaml
    from studiogan_amd.worker import Worker
    z = np.load(os.path.join(GOLDEN, "consistency.npz"))
    meta_c = json.load(open(os.path.join(GOLDEN, "consistency.json")))[tag]
    hp = meta_c["hp"]
    fix, meta = load_golden(meta_c["config"])
    y = meta["yaml"]
    G, D = build_from_yaml(y, False, dev)
    G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
    D.load_state_dict({k: v.to(dev) for k, v in sub(fix, "D_init/").items()}, strict=True)
    opt = hyper(y)
    bcr = hp.get("bcr_lambdas")
    w = Worker(G, D, opt["z_dim"], y["DATA"]["num_classes"], meta["batch"], hp.get("adv_loss", opt["adv_loss"]), opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"],
               d_updates_per_step=1, apply_diffaug=bool(hp.get("diffaug_policy")), apply_cr=hp.get("cr_lambda") is not None, cr_lambda=hp.get("cr_lambda", 0.0),
               apply_bcr=bcr is not None, real_lambda=(bcr or [0, 0])[0], fake_lambda=(bcr or [0, 0])[1], apply_zcr=hp.get("d_lambda") is not None,
               radius=hp.get("radius", 0.0), g_lambda=hp.get("g_lambda", 0.0), d_lambda=hp.get("d_lambda", 0.0),
               apply_fm=hp.get("fm_lambda") is not None, fm_lambda=hp.get("fm_lambda", 0.0),
               apply_apa=hp.get("apa_p") is not None, apa_initial_augment_p=hp.get("apa_p", 0.0), apa_target=None,
               apply_wc=hp.get("wc_bound") is not None, wc_bound=hp.get("wc_bound", 0.0),
               apply_ada=hp.get("ada_type") is not None, ada_aug_type=hp.get("ada_type", "bgc"), ada_initial_augment_p=hp.get("ada_p", 0.0), ada_target=None)
    if hp.get("diffaug_policy"):
        from studiogan_amd import diffaug as DA
        w.series_augment = lambda x: DA.apply_diffaug(x, hp["diffaug_policy"])
    ins = {k: v.to(dev) for k, v in sub(fix, "in/").items()}
    p = tag + "/"

    def draws_of(prefix, names):
        out = []
        for name in names:
            pre = f"{p}{prefix}/{name}/"
            keys = sorted((k for k in z.files if k.startswith(pre)), key=lambda k: int(k[len(pre):]))
            out += [torch.from_numpy(z[k]) for k in keys]
        return out
    C = Collector()
    zed = torch.from_numpy(z[p + "z_eps_d"]).to(dev) if p + "z_eps_d" in z.files else None
    with Replayed(draws_of("draw_d", ["apa", "series_real", "series_fake", "prl_real", "prl_fake", "ada"])):
        d_loss = w.train_discriminator(0, [(ins["real0"], ins["rl0"])], [(ins["z0"], ins["fl0"]) + ((zed,) if zed is not None else ())])
    C.check("d_loss", d_loss, torch.from_numpy(z[p + "d_loss"]), 2e-4)
    dmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith(p + "D_grad/"))
    for k, prm in D.named_parameters():
        C.check("D_grad/" + k, prm.grad, torch.from_numpy(z[p + "D_grad/" + k]), 1e-3, floor=1e-2 * dmax)
    if hp.get("wc_bound") is not None:
        # after Adam + the clip (src/worker.py:440-443,489-492): inside the band an element whose gradient is rounding noise may land one lr-kick apart
        b = hp["wc_bound"]
        for k, prm in D.named_parameters():
            assert float(prm.detach().abs().max()) <= b, k
            C.check("D_after/" + k, prm, torch.from_numpy(z[p + "D_after/" + k]), 1e-3, abs_ok=2.5 * opt["d_lr"])
    # the generator side of the fixture was taken on the networks as the discriminator side's FORWARDS left them (no optimiser step in between):
    # put the discriminator's weights back, keep the spectral-norm vectors / batch-norm statistics where the update's forwards left them
    with torch.no_grad():
        for k, prm in D.named_parameters():
            prm.copy_(fix["D_init/" + k].to(dev))
    zeg = torch.from_numpy(z[p + "z_eps_g"]).to(dev) if p + "z_eps_g" in z.files else None
    with Replayed(draws_of("draw_g", ["series_fake", "series_real_fm", "ada"])):
        g_loss = w.train_generator(0, [(ins["z1"], ins["fl1"]) + ((zeg,) if zeg is not None else ())], real_batches=[(ins["real1"], ins["rl1"])])
    C.check("g_loss", g_loss, torch.from_numpy(z[p + "g_loss"]), 1e-3)
    gmx = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith(p + "G_grad/"))
    for k, prm in G.named_parameters():
        C.check("G_grad/" + k, prm.grad, torch.from_numpy(z[p + "G_grad/" + k]), 2e-2, floor=1e-2 * gmx)
    C.finish()


CONSISTENCY_CASES = ["biggan32_diffaug_bcr_zcr", "sngan32_cr", "sngan32_diffaug", "sngan32_ls_fm_diffaug", "sngan32_apa_wc", "sngan32_ada"]


def apa_case(i, dev):
    """apply_apa_aug against the reference's output under the same draw (bit for bit: a select), and the accumulator of the heuristic"""
    from studiogan_amd import apa_aug, functional as F
    z = np.load(GOLD)
    p = f"apa/{i}/"
    real, fake = _t(z, p + "real").to(dev), _t(z, p + "fake").to(dev)
    with Replayed([_t(z, p + "coin")]):
        y = apa_aug.apply_apa_aug(real, fake, float(z[p + "p"]), dev)
    assert torch.equal(y.cpu(), _t(z, p + "y")), i
    # the real batch stays differentiable through the select (reference: fake * flag + real * (1 - flag); R1 on top of APA differentiates it twice, src/worker.py:274,379-381)
    rq = real.detach().clone().requires_grad_(True)
    with Replayed([_t(z, p + "coin")]):
        yq = apa_aug.apply_apa_aug(rq, fake, float(z[p + "p"]), dev)
    keep = (_t(z, p + "coin") >= float(z[p + "p"])).to(dev).float().reshape(-1, 1, 1, 1)
    wgt = torch.arange(yq.numel(), device=dev, dtype=torch.float32).reshape(yq.shape) / yq.numel()
    (g1,) = torch.autograd.grad((yq * yq * wgt).sum(), rq, create_graph=True)        # d/d real = 2 * y * wgt on the kept rows
    assert torch.equal(g1.detach(), (2 * yq.detach() * wgt) * keep), i
    (g2,) = torch.autograd.grad(g1.sum(), rq)                                        # second order: 2 * wgt on the kept rows
    assert torch.equal(g2, 2 * wgt * keep), i
    acc = torch.zeros(2, dtype=torch.float32, device=dev)
    lg = torch.tensor([0.5, -1.0, 0.0, 2.0, -0.1, 3.0, 1e-9], device=dev)
    F.sign_count_(acc, lg)
    F.sign_count_(acc, lg[:3])
    assert acc.tolist() == [float(torch.sign(lg).sum() + torch.sign(lg[:3]).sum()), 10.0]


def clamp_case(dev):
    """FusedAdam.clamp_ == p.data.clamp_(-b, b) on every parameter (reference src/worker.py:489-492), odd sizes included"""
    from studiogan_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    ps = [torch.nn.Parameter((torch.randn(shape, generator=g) * 0.05).to(dev)) for shape in ((7,), (3, 5, 3, 3), (1,), (129, 33))]
    ref = [q.detach().clone().clamp_(-0.03, 0.03) for q in ps]
    opt = FusedAdam(ps, lr=1e-3)
    opt.zero_grad()
    opt.clamp_(0.03)
    for q, r in zip(ps, ref):
        assert torch.equal(q.detach(), r)


def loss_case(kind, dev):
    """least-squares / logistic (/ vanilla) adversarial losses: value and gradients against the reference's functions (utils/losses.py:197-223)"""
    from studiogan_amd import losses
    z = np.load(GOLD)
    p = f"loss/{kind}/"
    r, f = _t(z, p + "real").to(dev).requires_grad_(True), _t(z, p + "fake").to(dev).requires_grad_(True)
    dl = losses.D_LOSSES[kind](r, f, DDP=False)
    dr, df = torch.autograd.grad(dl, [r, f], torch.tensor(1.3, device=dev))
    gl = losses.G_LOSSES[kind](f, DDP=False)
    (gf,) = torch.autograd.grad(gl, f, torch.tensor(1.3, device=dev))
    for name, a, b in (("d", dl, "d"), ("d dreal", dr, "d_dreal"), ("d dfake", df, "d_dfake"), ("g", gl, "g"), ("g dfake", gf, "g_dfake")):
        check(f"{kind} {name}", a.reshape(-1), _t(z, p + b).reshape(-1), 3e-6)


def fm_case(case, dev):
    from studiogan_amd import losses
    tag, shape = case
    z = np.load(GOLD)
    p = f"fm/{tag}/"
    hr, hf = _t(z, p + "real").to(dev).requires_grad_(True), _t(z, p + "fake").to(dev).requires_grad_(True)
    fm = losses.feature_matching_loss(hr.detach(), hf)
    (dh,) = torch.autograd.grad(fm, hf, torch.tensor(0.9, device=dev))
    check(f"fm {tag}", fm.reshape(-1), _t(z, p + "loss").reshape(-1), 3e-6)
    # the gradient is sign(column difference) / (B C): the signs agree wherever the difference is not within rounding distance of 0
    exp = _t(z, p + "dfake")
    flipped = (torch.sign(dh.detach().cpu()) != torch.sign(exp)).any(0)
    diff = (_t(z, p + "fake").mean(0) - _t(z, p + "real").mean(0)).abs()
    assert not bool(flipped.any()) or float(diff[flipped].max()) <= 1e-6, tag
    keep = ~flipped
    check(f"fm {tag} dfake", dh.detach().cpu()[:, keep], exp[:, keep], 1e-6)


LOSS_KINDS, FM_CASES = MGA.LOSS_KINDS, MGA.FM_CASES


def r1_through_diffaug_case(name, dev):
    """R1 on a real batch that reaches the discriminator THROUGH DiffAugment (reference src/worker.py:260-261,276-279,410-412 with src/utils/losses.py:355-361;
    the StyleGAN2-DiffAug family's combination): the create_graph pass runs through functional.AugmentFn / AugmentBwdFn inside the network's graph. Penalty and
    every parameter gradient against torch autograd's double backward over the oracle fed the same draws (fp32)."""
    from util import load_golden, sub, Collector
    from test_model_gpu import build_from_yaml
    from test_blocks_gpu import _split, _perturb
    from oracle import restate as O, make_golden as MG
    from studiogan_amd import losses as SL, diffaug as DA
    policy = "color,translation,cutout"
    fix, meta = load_golden(name)
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    P, B = _split(sub(fix, "D_init/"))
    _perturb(P, 9)
    _, D = build_from_yaml(y, False, dev)
    D.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    D.train()
    real, lab = fix["in/real0"].clone(), fix["in/rl0"]
    torch.manual_seed(77)
    draws = AR.draw_diffaug(tuple(real.shape), policy)
    dis = O.model_fns(ocfg)[1]
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    r_o, adv_o = O.r1_reg(lambda x, l, Pp, Bb: dis(AR.diffaug(x, policy, draws), l, Pp, Bb), real, lab, leaves, B)
    (10.0 * r_o + torch.mean(torch.relu(1.0 - adv_o))).backward()
    for prm in D.parameters():
        prm.grad = None
    xr = real.to(dev).requires_grad_(True)
    with Replayed(draws):
        out = D(DA.apply_diffaug(xr, policy), lab.to(dev))
    r = SL.cal_r1_reg(adv_output=out["adv_output"], images=xr, device=dev)
    (10.0 * r + SL.d_hinge(out["adv_output"], torch.full_like(out["adv_output"].detach(), -5.0))).backward()
    C = Collector()
    C.check("r1 through diffaug: penalty", r, r_o, 5e-4)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for k, prm in D.named_parameters():
        go = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        C.check("r1 through diffaug: grad " + k, prm.grad if prm.grad is not None else torch.zeros_like(prm), go, 1e-3, floor=1e-2 * gmax)
    C.finish()


# ---- adaptive discriminator augmentation against the reference's own outputs (tests/golden/ada.npz, oracle/make_golden_ada.py) ------------------------------
ADA_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ada.npz")


def ada_case(case, dev):
    """studiogan_amd.ada_aug.AdaAugment fed the draws the REAL reference's AdaAugment made: output and image gradient (fp32; 2e-5 of the tensor's range: the
    reference's grid coordinates come out of a batched matmul, the kernel's out of three fused multiply-adds)"""
    from oracle import make_golden_ada as MGD
    from studiogan_amd import ada_aug
    tag, pipe, shape, pr = case
    z = np.load(ADA_GOLD)
    pre = f"{tag}/"
    draws = []
    while pre + f"draw{len(draws)}" in z.files:
        draws.append(_t(z, pre + f"draw{len(draws)}"))
    aug = ada_aug.AdaAugment(**MGD.PIPES[pipe]).to(dev)
    aug.p.copy_(torch.as_tensor(pr))
    x = _t(z, pre + "x").to(dev).requires_grad_(True)
    with Replayed(draws) as R:
        y = aug(x)
    assert next(R.it, None) is None, "the mirror consumed fewer draws than the reference made"
    check(f"ada {tag} y", y, _t(z, pre + "y"), 2e-5)
    if float(np.abs(z[pre + "dx"]).max()) > 0:
        (dx,) = torch.autograd.grad(y, x, _t(z, pre + "gy").to(dev))
        check(f"ada {tag} dx", dx, _t(z, pre + "dx"), 2e-5)


def ada_adjoint_case(shape, dev, seed=0):
    """the three image-side operators against their adjoints at any size: <A u, v> == <u, A^T v> for reflect padding, the affine resampling (random
    rotations / scalings / shifts) and the colour matrix; run-to-run bit-identity of the gather-form backward"""
    from studiogan_amd import functional as F
    g = torch.Generator().manual_seed(seed)
    N, C, H, W = shape

    def adj(fn, bfn, u, v):
        Au, Atv = fn(u), bfn(v)
        lhs, rhs = float((Au.double() * v.double()).sum()), float((u.double() * Atv.double()).sum())
        scale = float(Au.double().norm() * v.double().norm()) + 1e-30
        assert abs(lhs - rhs) <= 2e-6 * scale, (lhs, rhs, scale)
        assert torch.equal(Atv, bfn(v))
    u = torch.randn(shape, generator=g).to(dev)
    l, r, t, b = min(5, W - 1), min(2, W - 1), min(3, H - 1), min(H - 1, 7)
    adj(lambda a: F.ReflectPad2dFn.apply(a, l, r, t, b), lambda a: F.ReflectPad2dBwdFn.apply(a, l, r, t, b), u,
        torch.randn((N, C, H + t + b, W + l + r), generator=g).to(dev))
    ang = (torch.rand(N, generator=g) * 2 - 1) * 3.14159
    sx, sy = torch.exp2(torch.randn(N, generator=g) * 0.3), torch.exp2(torch.randn(N, generator=g) * 0.3)
    theta = torch.stack([torch.stack([sx * torch.cos(ang), -sy * torch.sin(ang), torch.randn(N, generator=g) * 0.2], 1),
                         torch.stack([sx * torch.sin(ang), sy * torch.cos(ang), torch.randn(N, generator=g) * 0.2], 1)], 1).to(dev)
    Ho, Wo = H + 6, W + 4
    adj(lambda a: F.AffineSampleFn.apply(a, theta, Ho, Wo), lambda a: F.AffineSampleBwdFn.apply(a, theta, H, W, Ho, Wo), u,
        torch.randn((N, C, Ho, Wo), generator=g).to(dev))
    # the resampling against torch's own affine_grid + grid_sample on the CPU
    ref = torch.nn.functional.grid_sample(u.cpu(), torch.nn.functional.affine_grid(theta.cpu(), [N, C, Ho, Wo], align_corners=False), mode="bilinear",
                                          padding_mode="zeros", align_corners=False)
    check("affine_sample vs torch grid_sample", F.AffineSampleFn.apply(u, theta, Ho, Wo), ref, 2e-5)
    if C in (1, 3):
        M = torch.randn(N, 3, 4, generator=g).to(dev)
        M0 = M.clone()
        M0[:, :, 3] = 0
        adj(lambda a: F.ColorAffineFn.apply(a, M0), lambda a: F.ColorAffineFn.apply(a, M, True), u, torch.randn(shape, generator=g).to(dev))


def ada_filter_adjoint_case(shape, dev, seed=0):
    """sg_fir_reflect (both axes) and the cutout mask against their adjoints: <A u, v> == <u, A^T v>; the filter against F.pad(mode='reflect') + a grouped conv1d on the CPU"""
    from studiogan_amd import functional as F
    g = torch.Generator().manual_seed(seed)
    N, C, H, W = shape
    T = 43 if min(H, W) > 21 else 2 * ((min(H, W) - 1) // 2) - 1
    taps = torch.randn(N, T, generator=g) * 0.2
    u, v = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    for axis in (0, 1):
        Au = F.FirReflectFn.apply(u.to(dev), taps.to(dev), axis)
        Atv = F.FirReflectFn.apply(v.to(dev), taps.to(dev), axis, True)
        lhs, rhs = float((Au.double().cpu() * v.double()).sum()), float((u.double() * Atv.double().cpu()).sum())
        assert abs(lhs - rhs) <= 2e-6 * float(Au.double().norm() * v.double().norm() + 1e-30), (axis, lhs, rhs)
        p = T // 2
        pad = [p, p, 0, 0] if axis == 0 else [0, 0, p, p]
        xp = torch.nn.functional.pad(u.reshape(1, N * C, H, W), pad, mode="reflect")
        w = taps.repeat_interleave(C, 0).reshape(N * C, 1, 1, T) if axis == 0 else taps.repeat_interleave(C, 0).reshape(N * C, 1, T, 1)
        ref = torch.nn.functional.conv2d(xp, w, groups=N * C).reshape(N, C, H, W)
        check(f"fir_reflect axis {axis} vs torch", Au, ref, 5e-6)
    cut = torch.cat([torch.rand(N, 2, generator=g), torch.full((N, 2), 0.5)], 1)
    Au = F.NoiseCutoutFn.apply(u.to(dev), None, None, cut.to(dev))
    Atv = F.NoiseCutoutFn.apply(v.to(dev), None, None, cut.to(dev))
    assert abs(float((Au.double().cpu() * v.double()).sum()) - float((u.double() * Atv.double().cpu()).sum())) <= 1e-6 * float(u.double().norm() * v.double().norm())


# ---- InfoGAN against the reference's vectors (tests/golden/info.npz, oracle/make_golden_info.py) -----------------------------------------------------------
INFO_CASES = ["biggan32_info_cbn", "sngan32_info_concat", "sngan32_info_cbn", "bigdeep32_info_cbn", "dcgan32_info_cbn"]


def info_case(tag, dev):
    """studiogan_amd.worker.Worker(info_type=...) on the reference's InfoGAN networks (generator code injection "cBN" / "concat", discriminator Q heads): the
    discriminator update (loss, every gradient, the Q heads untouched by its optimiser step), the generator update (adversarial + information loss, the gradients
    of every generator parameter AND of the Q heads) and the Q heads after their Adam step with the generator's settings, against the REAL reference (fp32)."""
    import importlib
    import json
    import types
    from util import Collector, GOLDEN, hyper
    from studiogan_amd import ops
    from studiogan_amd.worker import Worker
    z = np.load(os.path.join(GOLDEN, "info.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "info.json")))[tag]
    y, B = meta["yaml"], meta["batch"]
    M, Dt = y["MODEL"], y["DATA"]
    MODEL = types.SimpleNamespace(info_type=M["info_type"], g_info_injection=M["g_info_injection"], info_num_discrete_c=M.get("info_num_discrete_c", "N/A"),
                                  info_dim_discrete_c=M.get("info_dim_discrete_c", "N/A"), info_num_conti_c=M.get("info_num_conti_c", "N/A"))
    bb = importlib.import_module("studiogan_amd.backbones." + M.get("backbone", "resnet"))
    MOD = ops.Modules(apply_g_sn=M.get("apply_g_sn", False), apply_d_sn=M.get("apply_d_sn", False), g_cond_mtd=M.get("g_cond_mtd", "W/O"),
                      backbone=M.get("backbone", "resnet"), g_info_injection=M["g_info_injection"])
    G = bb.Generator(M.get("z_dim", 128), M.get("g_shared_dim", "N/A"), Dt["img_size"], M.get("g_conv_dim", 64), M.get("apply_attn", False),
                     M.get("attn_g_loc", ["N/A"]), M.get("g_cond_mtd", "W/O"), Dt["num_classes"], "ortho", M.get("g_depth", "N/A"), False, MOD, MODEL).to(dev)
    D = bb.Discriminator(Dt["img_size"], M.get("d_conv_dim", 64), M.get("apply_d_sn", False), M.get("apply_attn", False), M.get("attn_d_loc", ["N/A"]),
                         M.get("d_cond_mtd", "W/O"), M.get("aux_cls_type", "W/O"), M.get("d_embed_dim", "N/A"), M.get("normalize_d_embed", False),
                         Dt["num_classes"], "ortho", M.get("d_depth", "N/A"), False, MOD, MODEL).to(dev)
    p = tag + "/"
    compact = bool(meta.get("compact"))
    if compact:          # initial state by formula (oracle/make_golden.py formula_state), large expected tensors as samples + norm
        from oracle import make_golden as MG
        G.load_state_dict({k: v.to(dev) for k, v in MG.formula_state({k: list(v.shape) for k, v in G.state_dict().items()}, meta["seeds"][0]).items()}, strict=True)
        D.load_state_dict({k: v.to(dev) for k, v in MG.formula_state({k: list(v.shape) for k, v in D.state_dict().items()}, meta["seeds"][1]).items()}, strict=True)
    else:
        G.load_state_dict({k[len(p) + 7:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith(p + "G_init/")}, strict=True)
        D.load_state_dict({k[len(p) + 7:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith(p + "D_init/")}, strict=True)
    d_init = {k: v.detach().clone() for k, v in D.named_parameters()}

    def exp(key):
        """expected tensor, or (samples, norm) of a large one in a compact fixture"""
        if key in z.files:
            return torch.from_numpy(z[key])
        from util import Sampled
        return Sampled(torch.from_numpy(z[key + "#s"]), torch.tensor([0.0, float(z[key + "#n"][0])], dtype=torch.float64))

    def amax(key):
        return float(np.abs(z[key] if key in z.files else z[key + "#s"]).max())
    opt = hyper(y)
    Ls = y.get("LOSS", {})
    w = Worker(G, D, opt["z_dim"], Dt["num_classes"], B, opt["adv_loss"], opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"], d_updates_per_step=1,
               info_type=M["info_type"], info_num_discrete_c=M.get("info_num_discrete_c", 0), info_dim_discrete_c=M.get("info_dim_discrete_c", 0),
               info_num_conti_c=M.get("info_num_conti_c", 0), infoGAN_loss_discrete_lambda=Ls.get("infoGAN_loss_discrete_lambda", 1.0),
               infoGAN_loss_conti_lambda=Ls.get("infoGAN_loss_conti_lambda", 1.0))
    ins = {k[len(p) + 3:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith(p + "in/")}

    def codes(side):
        return tuple(torch.from_numpy(z[p + f"code_{side}_{kind}"]).to(dev) if p + f"code_{side}_{kind}" in z.files else None for kind in ("disc", "conti"))
    C = Collector()
    info_names = [k for k, _ in D.named_parameters() if k.startswith("info_")]
    assert info_names
    d_loss = w.train_discriminator(0, [(ins["real0"], ins["rl0"])], [(ins["z0"], ins["fl0"], None, codes("d"))])
    C.check("d_loss", d_loss, torch.from_numpy(z[p + "d_loss"]), 2e-4)
    wide = compact           # full widths: ~1e6 ReLU units per layer, a handful within fp32 rounding of 0 -> l2 metric (tests/test_model_gpu.py)
    dmax = max(amax(p + "D_grad/" + k) for k, _ in D.named_parameters())
    for k, prm in D.named_parameters():
        C.check("D_grad/" + k, prm.grad if prm.grad is not None else torch.zeros_like(prm), exp(p + "D_grad/" + k), 3e-3 if wide else 1e-3, floor=1e-2 * dmax, l2=wide)
        if k in info_names:          # the Q heads are not the discriminator optimiser's to move (src/config.py:509-517): bit for bit where they were
            assert torch.equal(prm.detach(), d_init[k]), k
    # the generator side of the fixture ran on the discriminator AFTER its optimiser step: Adam's sign noise on near-zero gradients aside, take the reference's
    if not compact:
        with torch.no_grad():
            for k, prm in D.named_parameters():
                prm.copy_(torch.from_numpy(z[p + "D_after_d/" + k]).to(dev))
    g_loss = w.train_generator(0, [(ins["z1"], ins["fl1"], None, codes("g"))])
    C.check("g_loss", g_loss, torch.from_numpy(z[p + "g_loss"]), 5e-3 if wide else 1e-3)
    gmx = max(amax(p + "G_grad/" + k) for k, _ in G.named_parameters())
    for k, prm in G.named_parameters():
        C.check("G_grad/" + k, prm.grad, exp(p + "G_grad/" + k), 5e-2 if wide else 2e-2, floor=1e-2 * gmx, l2=wide)      # (wide: on OUR discriminator after its Adam step; measured 3e-3)
    for k, prm in D.named_parameters():
        if k in info_names:
            e = exp(p + "Q_grad/" + k)
            C.check("Q_grad/" + k, prm.grad, e, 2e-2 if wide else 1e-3, floor=1e-3 * amax(p + "Q_grad/" + k), l2=wide)
            C.check("Q_after_g/" + k, prm, exp(p + "Q_after_g/" + k), 1e-3, abs_ok=2.5 * opt["g_lr"])
    C.finish()


def freeze_d_case(name, dev, n_freeze=2):
    """RUN.freezeD = N (reference src/utils/misc.py:190-216, src/worker.py:219): the first N discriminator blocks receive no gradient and do not move; the loss and
    the gradients of every other parameter are those of the unfrozen update (the forward is the same): against the golden vectors of tests/golden/<name>.npz"""
    from util import load_golden, sub, hyper, Collector
    from test_model_gpu import build_from_yaml
    from studiogan_amd.worker import Worker
    fix, meta = load_golden(name)
    y = meta["yaml"]
    G, D = build_from_yaml(y, False, dev)
    G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
    D.load_state_dict({k: v.to(dev) for k, v in sub(fix, "D_init/").items()}, strict=True)
    opt = hyper(y)
    w = Worker(G, D, opt["z_dim"], y["DATA"]["num_classes"], meta["batch"], opt["adv_loss"], opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"],
               d_updates_per_step=1, freezeD=n_freeze)
    ins = {k: v.to(dev) for k, v in sub(fix, "in/").items()}
    exp = sub(fix, "exp/")
    before = {k: v.detach().clone() for k, v in D.named_parameters()}
    d_loss = w.train_discriminator(0, [(ins["real0"], ins["rl0"])], [(ins["z0"], ins["fl0"])])
    C = Collector()
    C.check("d_loss0", d_loss, exp["d_loss0"], 2e-4)
    dmax = max(float(v.abs().max()) for k, v in exp.items() if k.startswith("D_grad0/"))
    frozen = 0
    for k, prm in D.named_parameters():
        if any(f"blocks.{i}." in k for i in range(n_freeze)):
            frozen += 1
            assert not prm.requires_grad and torch.equal(prm.detach(), before[k]), k
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
        else:
            C.check("D_grad0/" + k, prm.grad, exp["D_grad0/" + k], 1e-3, floor=1e-2 * dmax)
            assert not torch.equal(prm.detach(), before[k]) or float(exp["D_grad0/" + k].abs().max()) == 0.0, k
    assert frozen > 0
    C.finish()


# ---- LOGAN's latent optimisation against the reference's vectors (tests/golden/logan.npz, oracle/make_golden_logan.py) ----------------------------------------
def logan_case(dev):
    """studiogan_amd.worker.Worker(apply_lo=True): one discriminator and one generator update of the LOGAN configuration (unconditional ResNet generator with batch
    norm, spectral-norm discriminator, uniform prior) whose losses back-propagate THROUGH d D(G(z)) / dz -- the create_graph pass runs through the generator's
    differentiable data-gradient operators (LinearDgradFn, ConvDgradFn, BNBwdFn, TanhGradFn) and the discriminator's: moved latents, transport cost, loss and every
    gradient against the REAL reference's double backward (fp32).

    Tolerances (round 6, measured on the MI355X: profiles/r06_logan_tie.txt): the fixture's discriminator-side latents put ONE ReLU input of the discriminator at
    2.5e-7 (1.2e-7 of its tensor's range; with ~1e6 ReLU inputs per pass no seed in 200 avoids such a unit). torch's fp32 on a CPU, fp64 and the CPU interpreter all
    land on one side of it, the MFMA's summation order on the other: d D(G(z)) / dz moves by 2.5e-3 and the discriminator's gradients by up to 5e-3 -- while the same
    kernels meet the fp64 oracle at 1e-6 on the latents scaled by 1.0001, 0.9999, 0.9 (tools/diag_logan2.py) and on the generator-side latents of this fixture. The
    discriminator-side bounds are therefore the generator side's (2e-2 of the largest gradient: one flipped unit of this 8-channel network, not rounding noise)."""
    import importlib
    import json
    import types
    from util import Collector, GOLDEN, hyper
    from studiogan_amd import ops
    from studiogan_amd.worker import Worker
    z = np.load(os.path.join(GOLDEN, "logan.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "logan.json")))
    y = meta["yaml"]
    M, Dt, Ls = y["MODEL"], y["DATA"], y["LOSS"]
    MODEL = types.SimpleNamespace(info_type="N/A", g_info_injection="N/A")
    bb = importlib.import_module("studiogan_amd.backbones.resnet")
    MOD = ops.Modules(apply_g_sn=False, apply_d_sn=True, g_cond_mtd="W/O", backbone="resnet")
    G = bb.Generator(M["z_dim"], "N/A", Dt["img_size"], M["g_conv_dim"], False, ["N/A"], "W/O", Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
    D = bb.Discriminator(Dt["img_size"], M["d_conv_dim"], True, False, ["N/A"], "W/O", "W/O", "N/A", False, Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
    G.load_state_dict({k[7:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("G_init/")}, strict=True)
    D.load_state_dict({k[7:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("D_init/")}, strict=True)
    opt = hyper(y)
    w = Worker(G, D, opt["z_dim"], Dt["num_classes"], y["OPTIMIZATION"]["batch_size"], "hinge", opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"],
               d_updates_per_step=1, apply_lo=True, lo_rate=Ls["lo_rate"], lo_steps4train=Ls["lo_steps4train"], lo_alpha=Ls["lo_alpha"], lo_beta=Ls["lo_beta"],
               lo_lambda=Ls["lo_lambda"], z_prior="uniform")
    ins = {k[3:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("in/")}
    d_init = {k: v.detach().clone() for k, v in D.named_parameters()}
    C = Collector()
    s0, s1 = meta["mask_seeds"]
    with Replayed([torch.from_numpy(z[f"mask_draw/{s0}"])]):
        d_loss = w.train_discriminator(0, [(ins["real0"], ins["rl0"])], [(ins["z0"], ins["fl0"])])
    C.check("d transport cost", w.trsp_cost, torch.from_numpy(z["d_trsp_cost"]), 1e-3)
    C.check("d_loss", d_loss, torch.from_numpy(z["d_loss"]), 2e-4)
    dmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("D_grad/"))
    for k, prm in D.named_parameters():
        C.check("D_grad/" + k, prm.grad, torch.from_numpy(z["D_grad/" + k]), 2e-2, floor=1e-2 * dmax)
    with torch.no_grad():           # the generator side of the fixture ran on the un-stepped discriminator
        for k, prm in D.named_parameters():
            prm.copy_(d_init[k])
    with Replayed([torch.from_numpy(z[f"mask_draw/{s1}"])]):
        g_loss = w.train_generator(0, [(ins["z1"], ins["fl1"])])
    C.check("g transport cost", w.trsp_cost, torch.from_numpy(z["g_trsp_cost"]), 1e-3)
    C.check("g_loss", g_loss, torch.from_numpy(z["g_loss"]), 1e-3)
    gmx = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("G_grad/"))
    for k, prm in G.named_parameters():
        C.check("G_grad/" + k, prm.grad, torch.from_numpy(z["G_grad/" + k]), 2e-2, floor=1e-2 * gmx)
    C.finish()


def logan_oracle_case(dev, scale):
    """The discriminator update of logan_case on the fixture's latents times `scale`, against the fp64 CPU ORACLE's double backward (oracle/restate.py networks +
    the five lines of src/utils/losses.py:278-298 restated on them) at rounding-level bounds: the tight counterpart of logan_case's discriminator side, on inputs
    whose ReLU inputs stay clear of zero on this hardware (measured: profiles/r06_logan_tie.txt -- 0.9, 0.9999, 1.0001 agree to 1e-6, the exact fixture latents do not)."""
    import importlib
    import json
    import types
    from util import Collector, GOLDEN, hyper
    from oracle import restate as O, make_golden as MG
    from studiogan_amd import ops
    from studiogan_amd.worker import Worker
    z = np.load(os.path.join(GOLDEN, "logan.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "logan.json")))
    y = meta["yaml"]
    M, Dt, Ls = y["MODEL"], y["DATA"], y["LOSS"]
    MODEL = types.SimpleNamespace(info_type="N/A", g_info_injection="N/A")
    bb = importlib.import_module("studiogan_amd.backbones.resnet")
    MOD = ops.Modules(apply_g_sn=False, apply_d_sn=True, g_cond_mtd="W/O", backbone="resnet")
    G = bb.Generator(M["z_dim"], "N/A", Dt["img_size"], M["g_conv_dim"], False, ["N/A"], "W/O", Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
    D = bb.Discriminator(Dt["img_size"], M["d_conv_dim"], True, False, ["N/A"], "W/O", "W/O", "N/A", False, Dt["num_classes"], "ortho", "N/A", False, MOD, MODEL).to(dev)
    gsd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G_init/")}
    dsd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("D_init/")}
    G.load_state_dict({k: v.to(dev) for k, v in gsd.items()}, strict=True)
    D.load_state_dict({k: v.to(dev) for k, v in dsd.items()}, strict=True)
    opt = hyper(y)
    w = Worker(G, D, opt["z_dim"], Dt["num_classes"], y["OPTIMIZATION"]["batch_size"], "hinge", opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"],
               d_updates_per_step=1, apply_lo=True, lo_rate=Ls["lo_rate"], lo_steps4train=Ls["lo_steps4train"], lo_alpha=Ls["lo_alpha"], lo_beta=Ls["lo_beta"],
               lo_lambda=Ls["lo_lambda"], z_prior="uniform")
    real, rl, fl = torch.from_numpy(z["in/real0"]), torch.from_numpy(z["in/rl0"]), torch.from_numpy(z["in/fl0"])
    z0 = torch.from_numpy(z["in/z0"]) * scale
    draw = torch.from_numpy(z[f"mask_draw/{meta['mask_seeds'][0]}"])
    # ---- oracle, fp64 ----
    gen, dis = O.model_fns(MG.oracle_cfg(y))
    pg, pd = {k for k, _ in G.named_parameters()}, {k for k, _ in D.named_parameters()}
    f64 = lambda v: v.clone().double() if v.is_floating_point() else v.clone()
    GP, GB = {k: f64(v) for k, v in gsd.items() if k in pg}, {k: f64(v) for k, v in gsd.items() if k not in pg}
    DP, DB = {k: f64(v).requires_grad_(True) for k, v in dsd.items() if k in pd}, {k: f64(v) for k, v in dsd.items() if k not in pd}
    zc = z0.double().requires_grad_(True)
    adv, _ = dis(gen(zc, fl, GP, GB, "untrack"), fl, DP, DB)
    zg = torch.autograd.grad(adv.sum(), zc, create_graph=True)[0]
    delta = Ls["lo_alpha"] * zg / (Ls["lo_beta"] + (zg.norm(2, dim=1) ** 2).unsqueeze(1))
    zs_o = torch.clamp(zc + (draw > 1 - Ls["lo_rate"]).double() * delta, -1.0, 1.0)
    cost_o = (delta.norm(2, dim=1) ** 2).mean()
    fake_o = gen(zs_o, fl, GP, GB, "untrack")
    rd, _ = dis(real.double(), rl, DP, DB)
    fd, _ = dis(fake_o, fl, DP, DB)
    loss_o = O.d_loss("hinge", rd, fd) + Ls["lo_lambda"] * cost_o
    grads_o = dict(zip(DP, torch.autograd.grad(loss_o, list(DP.values()))))
    # ---- product ----
    C = Collector()
    with Replayed([draw]):
        d_loss = w.train_discriminator(0, [(real.to(dev), rl.to(dev))], [(z0.to(dev), fl.to(dev))])
    C.check("d transport cost", w.trsp_cost, cost_o.detach(), 2e-5)
    C.check("d_loss", d_loss, loss_o.detach(), 2e-5)
    dmax = max(float(g.abs().max()) for g in grads_o.values())
    for k, prm in D.named_parameters():
        C.check("D_grad/" + k, prm.grad, grads_o[k], 1e-4, floor=1e-2 * dmax)
    C.finish()


def r1_with_heads_case(name, dev):
    """R1 on a BigGAN discriminator with attention AND a classifier-style head (reference configs/*/MDGAN.yaml: `apply_r1_reg` with the multi-discriminator head; the
    adversarial logit runs through linear1 as functional.LinearFn -> LinearDgradFn in the create_graph pass, then the per-class gather): penalty and every
    parameter gradient against torch autograd's double backward over the oracle (networks of tests/golden/heads.npz)."""
    import json
    from util import Collector, GOLDEN
    from test_model_gpu import build_from_yaml
    from oracle import restate as O, make_golden as MG
    from studiogan_amd import losses as SL
    z = np.load(os.path.join(GOLDEN, "heads.npz"))
    c = json.load(open(os.path.join(GOLDEN, "heads.json")))["cases"][name]
    y = c["yaml"]
    _, D = build_from_yaml(y, False, dev)
    sd = {k[len(name) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/P/") or k.startswith(name + "/B/")}
    D.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    D.train()
    P = {k[len(name) + 3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(name + "/P/")}
    B = {k[len(name) + 3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(name + "/B/")}
    for k in P:
        if k.endswith("sigma"):
            P[k] = torch.full_like(P[k], 0.6)          # attention switched on
    with torch.no_grad():
        for k, prm in D.named_parameters():
            prm.copy_(P[k].to(dev))
    real, rl = torch.from_numpy(z[name + "/in/real"]), torch.from_numpy(z[name + "/in/rl"])
    ocfg = MG.oracle_cfg(y)
    dis = O.model_fns(ocfg)[1]
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}

    def dis_adv(x, l, Pp, Bb):
        adv, h = dis(x, l, Pp, Bb)
        return O.d_heads(adv, h, l, Pp, Bb, ocfg)["adv_output"], h
    r_o, adv_o = O.r1_reg(dis_adv, real, rl, leaves, B)
    (10.0 * r_o + torch.mean(torch.relu(1.0 - adv_o))).backward()
    for prm in D.parameters():
        prm.grad = None
    xr = real.to(dev).requires_grad_(True)
    out = D(xr, rl.to(dev))
    r = SL.cal_r1_reg(adv_output=out["adv_output"], images=xr, device=dev)
    (10.0 * r + SL.d_hinge(out["adv_output"], torch.full_like(out["adv_output"].detach(), -5.0))).backward()
    C = Collector()
    C.check(f"r1 [{name}] penalty", r, r_o, 5e-4)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for k, prm in D.named_parameters():
        go = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        C.check(f"r1 [{name}] grad " + k, prm.grad if prm.grad is not None else torch.zeros_like(prm), go, 1e-3, floor=1e-2 * gmax)
    C.finish()


# ---- the evaluation-side generator preparation against the reference's vectors (tests/golden/standing.npz, oracle/make_golden_standing.py) -----------------------
STANDING_CASES = ["sngan", "biggan", "resgan"]


def standing_case(name, dev):
    """studiogan_amd.worker.GeneratorController.prepare_generator (reference src/utils/misc.py:63-107) in its three modes -- standing statistics (reset, then five
    training-mode forwards over batches of 1..7 latents: the draws replayed from the fixture), batch statistics, plain evaluation -- on a generator that has trained a
    little: every buffer afterwards (batch-norm running statistics and counters, spectral-norm vectors), the training flags of the leaf modules, and the evaluation-mode
    image of fixed latents, against what the REAL reference left behind (fp32)."""
    import json
    import random
    from util import Collector, GOLDEN
    from studiogan_amd import config_map as CM
    from studiogan_amd.worker import GeneratorController
    z = np.load(os.path.join(GOLDEN, "standing.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "standing.json")))
    c, pre = meta["cases"][name], name + "/"
    y = c["yaml"]
    seed = meta["seed"] + 20 + STANDING_CASES.index(name)
    draws = []
    while pre + f"draw{len(draws)}" in z.files:
        draws.append(torch.from_numpy(z[pre + f"draw{len(draws)}"]))
    C = Collector()
    for mode in ("standing", "batch", "plain"):
        G, _, _ = CM.build(y, dev)
        G.load_state_dict({k[len(pre) + 5:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith(pre + "init/")}, strict=True)
        ctl = GeneratorController(G, batch_statistics=mode == "batch", standing_statistics=mode == "standing", standing_max_batch=meta["max_batch"],
                                  standing_step=meta["steps"], z_dim=y["MODEL"]["z_dim"], num_classes=y["DATA"]["num_classes"], device=dev)
        random.seed(seed)
        with Replayed(draws), torch.no_grad():
            G, _, _ = ctl.prepare_generator()
        flags = sorted({(type(m).__name__.replace("SnConv2d", "Conv2d").replace("SnLinear", "Linear").replace("SnEmbedding", "Embedding"), m.training)
                        for m in G.modules() if isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding))})
        want = sorted({(a, b) for a, b in c[mode + "_training_flags"] if a in ("BatchNorm2d", "Conv2d", "Linear", "Embedding")})
        got = sorted({("BatchNorm2d" if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) else "Conv2d" if isinstance(m, torch.nn.Conv2d) else
                       "Linear" if isinstance(m, torch.nn.Linear) else "Embedding", m.training)
                      for m in G.modules() if isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding))})
        assert got == want, (name, mode, got, want, flags)
        with torch.no_grad():
            img = G(torch.from_numpy(z[pre + "z_eval"]).to(dev), torch.from_numpy(z[pre + "y_eval"]).to(dev), eval=True)
        C.check(f"{name} {mode} image", img, torch.from_numpy(z[pre + mode + "/image"]), 2e-4)
        bufs = dict(G.named_buffers())
        n = 0
        for k in z.files:
            if k.startswith(pre + mode + "/final/"):
                kk = k[len(pre + mode + "/final/"):]
                ref = torch.from_numpy(z[k])
                if ref.dtype == torch.int64:
                    assert int(bufs[kk]) == int(ref), (name, mode, kk, int(bufs[kk]), int(ref))
                else:
                    C.check(f"{name} {mode} {kk}", bufs[kk], ref, 2e-4, floor=1e-3)
                n += 1
        assert n > 0
    C.finish()


# ---- one training step of a reference configuration file against what the reference's OWN worker produced (tests/golden/config_steps.npz, tools/config_worker_parity_emulated.py --emit) ---
class ReplayedAll:
    """torch.rand / randn / randint / FloatTensor(...).uniform_() hand out the draws the reference's worker consumed, in its order, on whatever device is asked for
    (shape-checked; both call forms of torch.randint)"""

    def __init__(self, draws):
        self.it = iter(draws)

    def __enter__(self):
        self.saved = (torch.rand, torch.randn, torch.randint, torch.FloatTensor)
        it = self.it

        def nxt(shape, dev):
            d = next(it)
            if len(shape) == 1 and isinstance(shape[0], (list, tuple, torch.Size)):
                shape = tuple(shape[0])
            assert tuple(d.shape) == tuple(shape), (tuple(d.shape), tuple(shape))
            return d.to(dev) if dev is not None else d

        def rand(*size, dtype=None, device=None, **kw):
            return nxt(size, device)

        def randint(*a, size=None, device=None, low=None, high=None, **kw):
            if size is None:
                size = next(x for x in a if isinstance(x, (list, tuple, torch.Size)))
            return nxt(tuple(size), device)

        class FT:
            def __init__(self, *size):
                self.size = size

            def uniform_(self, a=0.0, b=1.0):
                return nxt(self.size, None)
        torch.rand, torch.randn, torch.randint, torch.FloatTensor = rand, rand, randint, FT
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn, torch.randint, torch.FloatTensor = self.saved
        if a[0] is None:
            assert next(self.it, None) is None, "the step consumed fewer draws than the reference's"


def config_step_names(fixture="config_steps"):
    import json
    from util import GOLDEN
    p = os.path.join(GOLDEN, fixture + ".json")
    return sorted(json.load(open(p))["cases"]) if os.path.exists(p) else []


def config_step_case(name, dev, fixture="config_steps"):
    """One training step (two discriminator updates with Adam in between, one generator update) of the reference configuration file `name` -- networks at width 8 built
    through config_map.build under torch.manual_seed(0), which reproduces the reference's initialisation bit for bit -- fed the draws the reference's UNMODIFIED
    WORKER.train_discriminator / train_generator consumed when it ran this file on the CPU (latents, labels, InfoGAN codes, augmentation and gradient-penalty draws: recorded
    in call order by tools/config_worker_parity_emulated.py --emit): the losses of the last updates and the l2 norms of the two networks' gradients against the reference's.
    Rows the tool found ill-conditioned (a ReLU near-tie at this input; full-width DCGAN's batch norm over four samples) are held to their losses only."""
    import json
    from util import GOLDEN
    from studiogan_amd import config_map as CM
    from studiogan_amd import worker as SW
    meta = json.load(open(os.path.join(GOLDEN, fixture + ".json")))
    z = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    c = meta["cases"][name]
    y, batch, n_d = c["yaml"], meta["batch"], meta["n_d"]
    draws = [torch.from_numpy(z[f"{name}/draw{i}"]) for i in range(c["draws"])]
    torch.manual_seed(0)
    G, D, w = CM.build(y, dev)
    kw = CM.worker_kwargs(y)
    S, nc = (y.get("DATA") or {}).get("img_size", 32), kw["num_classes"]
    g = torch.Generator().manual_seed(11)
    baskets = [(torch.randint(0, 256, (n_d * batch, 3, S, S), generator=g).float() / 127.5 - 1.0, torch.randint(0, nc, (n_d * batch,), generator=g)) for _ in range(2)]
    reals = [(baskets[0][0][i * batch:(i + 1) * batch].to(dev), baskets[0][1][i * batch:(i + 1) * batch].to(dev)) for i in range(n_d)]
    fm = [(baskets[1][0][:batch].to(dev), baskets[1][1][:batch].to(dev))] if kw["apply_fm"] else None
    with ReplayedAll(draws):          # (worker.sample_zy draws labels first, then the latents: the reference's order, src/utils/sample.py:69-76)
        d_loss = w.train_discriminator(1, reals)
        dn = torch.stack([p.grad.double().norm() for p in D.parameters() if p.grad is not None]).norm()
        g_loss = w.train_generator(1, real_batches=fm)
        gn = torch.stack([p.grad.double().norm() for p in G.parameters() if p.grad is not None]).norm()
    rel_ = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-3)
    e = {"d_loss": rel_(d_loss.detach(), z[f"{name}/d_loss"]), "g_loss": rel_(g_loss.detach(), z[f"{name}/g_loss"]),
         "d_grad_norm": rel_(dn, z[f"{name}/d_grad_norm"]), "g_grad_norm": rel_(gn, z[f"{name}/g_grad_norm"])}
    print(name, {k: f"{v:.1e}" for k, v in e.items()}, "ill-conditioned" if c["ill_conditioned"] else "")
    if c["ill_conditioned"]:
        assert e["d_loss"] <= 5e-2 and e["g_loss"] <= 5e-2, (name, e)
    else:
        # (gradient norms: 3e-2 -- another implementation's rounding can put a different ReLU unit on the other side of zero: tools/config_worker_parity_emulated.py measures
        # what one such unit does to a whole-network norm: <= 1e-2; the interpreter's values are <= 1.6e-3)
        assert e["d_loss"] <= 2e-3 and e["g_loss"] <= 2e-3 and e["d_grad_norm"] <= 3e-2 and e["g_grad_norm"] <= 3e-2, (name, e)
