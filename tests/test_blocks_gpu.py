"""GPU: whole-network forward/backward of every backbone's G and D against the CPU oracle with NON-initial parameters
(attention gate sigma != 0, random biases) -- exercises every backward path including the ones that are dead at
initialisation (attention branch, gradient w.r.t. the input image, the identity-skip block)."""
import json
import os

import pytest
import torch

from util import Collector, load_golden, sub
from oracle import make_golden as MG
from oracle import restate as O
from test_model_gpu import build_from_yaml

pytestmark = pytest.mark.gpu


def _split(init):
    isb = lambda k: any(s in k for s in ("weight_u", "weight_v", "running_", "num_batches"))
    P = {k: v.clone() for k, v in init.items() if not isb(k)}
    B = {k: v.clone() for k, v in init.items() if isb(k)}
    return P, B


def _perturb(P, seed):
    g = torch.Generator().manual_seed(seed)
    for k in P:
        if k.endswith("sigma"):
            P[k] = torch.full_like(P[k], 0.6)
        elif k.endswith(".bias") and P[k].dim() == 1:
            P[k] = 0.1 * torch.randn(P[k].shape, generator=g)
        elif k.endswith("bn4.weight"):
            P[k] = 1 + 0.2 * torch.randn(P[k].shape, generator=g)


NAMES = ["biggan32", "sngan32", "resgan32", "dcgan32", "sndcgan32", "bigdeep32", "bigdeepsg32"]


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_discriminator_fwd_bwd(sg, name, mixed):
    discriminator_fwd_bwd(name, mixed)


def discriminator_fwd_bwd(name, mixed):
    dev = torch.device("cuda:0")
    fix, meta = load_golden(name)
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    P, B = _split(sub(fix, "D_init/"))
    _perturb(P, 3)
    _, D = build_from_yaml(y, mixed, dev)
    D.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    D.train()
    x = fix["in/real0"].clone()
    lab = fix["in/rl0"]
    gadv = torch.tensor([0.3, -1.0, 0.7, 0.5, -0.2, 0.9, -0.6, 0.1]).repeat((x.shape[0] + 7) // 8)[:x.shape[0]]
    # oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xo = x.clone().requires_grad_(True)
    adv_o, h_o = O.model_fns(ocfg)[1](xo, lab, leaves, B)
    (adv_o * gadv).sum().backward()
    # HIP path
    xd = x.to(dev).requires_grad_(True)
    for p in D.parameters():
        p.grad = None
    out = D(xd, lab.to(dev))
    (out["adv_output"] * gadv.to(dev)).sum().backward()
    torch.cuda.synchronize()
    C = Collector()
    t = 2e-4 if not mixed else 5e-2
    # bf16 gradients: every ReLU layer flips the mask of ~0.2-0.3 % of its units under 2^-9 activation rounding, i.e. a
    # ~sqrt(f) ~ 4-5 % L2 perturbation per layer that accumulates with depth (measured: 0.5 % at the last layer -> 12 %
    # at the first); it is unbiased noise, identical in kind to fp16 autocast in the reference. Relative-L2 <= 25 %.
    tg = 4e-4 if not mixed else 0.25
    wide = bool(meta.get("compact"))     # full DCGAN widths: ~1e6 ReLU units per layer, a handful sit within fp32 rounding of 0
    if wide and not mixed:
        tg = 6e-3
    l2 = mixed or wide
    C.check("D adv", out["adv_output"], adv_o, t)
    C.check("D h", out["h"], h_o, t)
    C.check("D dx (input image gradient)", xd.grad, xo.grad, tg, l2=l2)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values())
    for k, p in D.named_parameters():
        C.check("D grad " + k, p.grad, leaves[k].grad, tg, floor=1e-2 * gmax, l2=l2)  # 1e-2: bias gradients in front of a BN are analytically 0 (pure cancellation noise)
    for k, b in D.named_buffers():
        C.check("D buf " + k, b, B[k], t)
    C.finish()


@pytest.mark.parametrize("mixed", [False, True])
def test_attention_input_gradient_chain_matches_autograd_sum(sg, mixed):
    """functional.GradLink(chain=True) in ops.SelfAttention: the four gradients the attention input receives (theta / phi / g / residual) are
    summed inside the three 1x1 data-gradient launches. Same network, same inputs, SG_GRAD_LINK on vs off: every gradient agrees (fp32: to
    rounding of the summation order; bf16: one bf16 rounding per partial sum that autograd's add<bf16> makes as well), and with the link on
    the three data-gradient launches of the attention really carry a residual."""
    from studiogan_amd import functional as F
    dev = torch.device("cuda:0")
    fix, meta = load_golden("biggan32")
    y = meta["yaml"]
    P, B = _split(sub(fix, "D_init/"))
    _perturb(P, 3)
    _, D = build_from_yaml(y, mixed, dev)
    D.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    D.train()
    x, lab = fix["in/real0"].clone(), fix["in/rl0"]
    sd = {k: v.clone() for k, v in D.state_dict().items()}
    got, with_res = [], []
    FC = F.conv                    # (the operator-family module whose functions look `_conv_dgrad` up: round 6's split of functional.py)
    orig, keep = FC._conv_dgrad, F._GRAD_LINK[0]
    try:
        for on in (False, True):
            n = [0]

            def counting(dy, xx, rt, slot, cfg, res=None):
                n[0] += (res is not None) and cfg.R == 1 and not cfg.in_relu
                return orig(dy, xx, rt, slot, cfg, res=res)
            FC._conv_dgrad, F._GRAD_LINK[0] = counting, on
            D.load_state_dict(sd)      # same power-iteration state for both passes
            for p in D.parameters():
                p.grad = None
            xd = x.to(dev).requires_grad_(True)
            D(xd, lab.to(dev))["adv_output"].sum().backward()
            torch.cuda.synchronize()
            got.append({"dx": xd.grad.float().cpu(), **{k: p.grad.float().cpu() for k, p in D.named_parameters()}})
            with_res.append(n[0])
    finally:
        FC._conv_dgrad, F._GRAD_LINK[0] = orig, keep
    assert with_res == [0, 3], with_res
    C = Collector()
    for k in got[0]:
        C.check("chain vs autograd sum: " + k, got[1][k], got[0][k], 1e-5 if not mixed else 2e-2, l2=True)
    C.finish()


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("bn_mode", ["track", "untrack", "eval"])
@pytest.mark.parametrize("name", NAMES)
def test_generator_fwd_bwd(sg, name, mixed, bn_mode):
    generator_fwd_bwd(name, mixed, bn_mode)


def generator_fwd_bwd(name, mixed, bn_mode):
    from studiogan_amd import worker as W
    dev = torch.device("cuda:0")
    fix, meta = load_golden(name)
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    P, B = _split(sub(fix, "G_init/"))
    _perturb(P, 4)
    for k in B:
        if "running_mean" in k:
            B[k] = 0.1 * torch.randn(B[k].shape, generator=torch.Generator().manual_seed(9))
        if "running_var" in k:
            B[k] = 0.5 + torch.rand(B[k].shape, generator=torch.Generator().manual_seed(10))
    G, _ = build_from_yaml(y, mixed, dev)
    G.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    if bn_mode == "eval":
        G.eval()
        G.apply(W.set_deterministic_op_trainable)   # reference utils/misc.py:254-262: SN layers keep iterating in eval
    else:
        G.train()
        G.apply(W.track_bn_statistics if bn_mode == "track" else W.untrack_bn_statistics)
    z, lab = fix["in/z0"], fix["in/fl0"]
    S = y["DATA"]["img_size"]
    gimg = torch.randn(z.shape[0], 3, S, S, generator=torch.Generator().manual_seed(11))
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    img_o = O.model_fns(ocfg)[0](z, lab, leaves, B, bn_mode=bn_mode)
    (img_o * gimg).sum().backward()
    for p in G.parameters():
        p.grad = None
    img = G(z.to(dev), lab.to(dev))
    (img * gimg.to(dev)).sum().backward()
    torch.cuda.synchronize()
    C = Collector()
    t = 2e-4 if not mixed else 5e-2
    # bf16 gradients: every ReLU layer flips the mask of ~0.2-0.3 % of its units under 2^-9 activation rounding, i.e. a
    # ~sqrt(f) ~ 4-5 % L2 perturbation per layer that accumulates with depth (measured: 0.5 % at the last layer -> 12 %
    # at the first); it is unbiased noise, identical in kind to fp16 autocast in the reference. Relative-L2 <= 25 %.
    tg = 4e-4 if not mixed else 0.25
    wide = bool(meta.get("compact"))
    if wide and not mixed:
        tg = 6e-3
    if mixed and name in ("bigdeep32", "bigdeepsg32"):
        # 48 cBN+ReLU layers at a width-8 bottleneck: against the FP32 oracle the bf16 mask-flip noise compounds to 30-60 % --
        # only a sanity bound here; the bf16 parity of this network is asserted against the emulating oracle (4-10 %)
        t, tg = 0.15, 0.9
    l2 = mixed or wide
    C.check(f"G img [{bn_mode}]", img, img_o, t)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values())
    deep_bf16 = mixed and name in ("bigdeep32", "bigdeepsg32")
    for k, p in G.named_parameters():
        # single scalars (the attention gate sigma: one dot product of cancelling terms) have no averaging over the mask-flip noise of
        # the deep bf16 nets against the FP32 oracle (measured 1.6 on bigdeepsg32; 2e-2 against the emulating oracle): finite-ness only
        tk = 4.0 if (deep_bf16 and p.numel() < 16) else tg
        C.check("G grad " + k, p.grad, leaves[k].grad, tk, floor=1e-2 * gmax, l2=l2)  # 1e-2: bias gradients in front of a BN are analytically 0 (pure cancellation noise)
    for k, b in G.named_buffers():
        if "_ones" not in k:
            C.check("G buf " + k, b, B[k], t)
    C.finish()


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("name", ["wgangp32", "sngp32", "resgan32", "sngan32", "dcgan32", "sndcgan32", "biggan32"])
def test_gradient_penalty_double_backward(sg, name, mixed):
    """WGAN-GP penalty (reference utils/losses.py:301-316) and its gradient w.r.t. every discriminator parameter -- the
    double backward through conv (stride 1 / 2, fused ReLU / pooling), batch norm with batch statistics, pooling, the
    projection head, spectral norm and (biggan32) self-attention -- against torch autograd's own double backward over the CPU oracle."""
    gradient_penalty_case(name, mixed, torch.device("cuda:0"))


def gradient_penalty_case(name, mixed, dev):
    from studiogan_amd import losses as SL
    fix, meta = load_golden(name)
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    P, B = _split(sub(fix, "D_init/"))
    _perturb(P, 5)
    _, D = build_from_yaml(y, mixed, dev)
    D.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    D.train()
    real, lab = fix["in/real0"].clone(), fix["in/rl0"]
    fake = fix["in/real1"].flip(0).clone() * 0.7
    alpha = MG.gp_alpha(meta["seed"], 0, real.shape[0])
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    gp_o = O.grad_penalty(O.model_fns(ocfg)[1], real, lab, fake, leaves, B, alpha)
    gp_o.backward()
    for p in D.parameters():
        p.grad = None
    torch.manual_seed(meta["seed"] + MG.GP_SEED)
    gp = SL.cal_grad_penalty(real.to(dev), lab.to(dev), fake.to(dev), D, dev)
    gp.backward()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    C = Collector()
    wide = bool(meta.get("compact"))
    t = 5e-4 if not mixed else 5e-2
    tg = (6e-3 if wide else 1e-3) if not mixed else 0.3
    l2 = mixed or wide
    C.check("penalty", gp, gp_o, t)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for k, p in D.named_parameters():
        go = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        C.check("gp grad " + k, p.grad if p.grad is not None else torch.zeros_like(p), go, tg, floor=(5e-2 if mixed else 1e-2) * gmax, l2=l2)
    for k, b in D.named_buffers():
        C.check("D buf " + k, b, B[k], t)
    C.finish()


@pytest.mark.parametrize("kind", ["r1", "maxgp"])
@pytest.mark.parametrize("name", ["wgangp32", "sngp32", "sngan32", "biggan32", "bigdeep32", "bigdeepsg32", "sndcgan32", "resgan32"])
def test_r1_and_maxgp_double_backward(sg, name, kind):
    """R1 (reference utils/losses.py:355-361, taken through the SAME real-batch forward that feeds the adversarial loss,
    src/worker.py:260-261,410-412) and the max-gradient penalty (:338-352): value and gradient w.r.t. every discriminator
    parameter against torch autograd's double backward over the CPU oracle (fp32). biggan32: through SelfAttention (spectral-normed theta / phi / g /
    output convolutions, max-pooling, softmax, the learnt gain sigma = 0.6) -- the create_graph pass re-evaluates the block from differentiable primitives
    (functional.AttnPooledFn.backward); bigdeep32 / bigdeepsg32: attention + the bottleneck blocks' channel-concat skip (functional.CatConvDgradFn);
    sndcgan32: strided 4x4 convolutions under spectral norm; resgan32: batch norm in the discriminator."""
    r1_maxgp_case(name, kind, torch.device("cuda:0"))


def r1_maxgp_case(name, kind, dev):
    """dev: the GPU, or the CPU when the package is bound to the interpreted library (tests/test_aug_cpu.py)"""
    from studiogan_amd import losses as SL
    fix, meta = load_golden(name)
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    P, B = _split(sub(fix, "D_init/"))
    _perturb(P, 6)
    _, D = build_from_yaml(y, False, dev)
    D.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    D.train()
    real, lab = fix["in/real0"].clone(), fix["in/rl0"]
    fake = fix["in/real1"].flip(0).clone() * 0.7
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    dis = O.model_fns(ocfg)[1]
    for p in D.parameters():
        p.grad = None
    if kind == "r1":
        r_o, adv_o = O.r1_reg(dis, real, lab, leaves, B)
        (10.0 * r_o + torch.mean(torch.relu(1.0 - adv_o))).backward()        # hinge on the real half + lambda * R1, one backward
        xr = real.to(dev).requires_grad_(True)
        out = D(xr, lab.to(dev))
        r = SL.cal_r1_reg(adv_output=out["adv_output"], images=xr, device=dev)
        (10.0 * r + SL.d_hinge(out["adv_output"], torch.full_like(out["adv_output"].detach(), -5.0))).backward()   # fake half: relu(1 - 5) = 0
    else:
        alpha = MG.gp_alpha(meta["seed"], 0, real.shape[0])
        r_o = O.maxgrad_penalty(dis, real, lab, fake, leaves, B, alpha)
        r_o.backward()
        torch.manual_seed(meta["seed"] + MG.GP_SEED)
        r = SL.cal_maxgrad_penalty(real.to(dev), lab.to(dev), fake.to(dev), D, dev)
        r.backward()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    C = Collector()
    C.check(kind + " penalty", r, r_o, 5e-4)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for k, p in D.named_parameters():
        go = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        C.check(kind + " grad " + k, p.grad if p.grad is not None else torch.zeros_like(p), go, 1e-3, floor=1e-2 * gmax)
    C.finish()


def test_dra_penalty_lecam_and_uint8_input(sg):
    """DRAGAN penalty (reference utils/losses.py:319-335), LeCam regulariser (:262-265 + utils/ops.py:106-133) and
    the uint8 dataset format as discriminator input (data_util.py:92-94,102-142), against the CPU oracle."""
    from studiogan_amd import losses as SL, ops, functional as F
    dev = torch.device("cuda:0")
    fix, meta = load_golden("sngp32")
    y = meta["yaml"]
    ocfg = MG.oracle_cfg(y)
    P, B = _split(sub(fix, "D_init/"))
    _perturb(P, 8)
    _, D = build_from_yaml(y, False, dev)
    D.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
    D.train()
    real, lab = fix["in/real0"].clone(), fix["in/rl0"]
    dis = O.model_fns(ocfg)[1]
    C = Collector()
    # -- DRA: same host draws on both sides (torch.rand(B,1,1,1) then torch.rand(real.size()), losses.py:321,325)
    torch.manual_seed(77)
    alpha, noise = torch.rand(real.shape[0], 1, 1, 1), torch.rand(real.size())
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    d_o = O.dra_penalty(dis, real, lab, leaves, B, alpha, noise)
    d_o.backward()
    for p in D.parameters():
        p.grad = None
    torch.manual_seed(77)
    d = SL.cal_dra_penalty(real.to(dev), lab.to(dev), D, dev)
    d.backward()
    torch.cuda.synchronize()
    C.check("dra penalty", d, d_o, 5e-4)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for k, p in D.named_parameters():
        go = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        C.check("dra grad " + k, p.grad if p.grad is not None else torch.zeros_like(p), go, 1e-3, floor=1e-2 * gmax)
    # -- LeCam: value and gradient w.r.t. the logits; the EMA class follows the reference's update rule
    g = torch.Generator().manual_seed(3)
    lr_, lf_ = torch.randn(37, generator=g), torch.randn(37, generator=g)
    ema = ops.LeCamEMA(init=0.0, decay=0.9, start_iter=2)
    ema.update(0.5, "D_real", 0); ema.update(-0.25, "D_fake", 0)          # itr < start_iter: decay 0 -> takes the value
    ema.update(1.0, "D_real", 5); ema.update(0.0, "D_fake", 5)
    assert abs(ema.D_real - (0.5 * 0.9 + 0.1)) < 1e-12 and abs(ema.D_fake - (-0.25 * 0.9)) < 1e-12
    ro, fo = lr_.clone().requires_grad_(True), lf_.clone().requires_grad_(True)
    l_o = O.lecam_reg(ro, fo, ema.D_real, ema.D_fake)
    l_o.backward()
    rd, fd = lr_.to(dev).requires_grad_(True), lf_.to(dev).requires_grad_(True)
    l = SL.lecam_reg(rd, fd, ema)
    l.backward()
    C.check("lecam", l, l_o, 1e-6)
    C.check("lecam d_real", rd.grad, ro.grad, 1e-6)
    C.check("lecam d_fake", fd.grad, fo.grad, 1e-6)
    # -- uint8 input: bit-identical to converting the host-normalised fp32 image, with and without flips; D accepts it directly
    xu = torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8)
    flip = torch.tensor([1, 0, 0, 1], dtype=torch.uint8)
    for fl in (None, flip):
        xn = O.uint8_to_normalized(xu, fl)
        for dt in (torch.float32, torch.bfloat16):
            a = F.u8_to_nhwc(xu.to(dev), dt, 8, None if fl is None else fl.to(dev))
            b = ops.to_nhwc(xn.to(dev), dt, 8)
            assert torch.equal(a.cpu(), b.cpu()), f"uint8 input path must be bit-exact ({dt}, flip={fl is not None})"
    D.eval()      # no power iteration between the two forwards
    with torch.no_grad():
        o1 = D(xu.to(dev), lab.to(dev))["adv_output"]
        o2 = D(O.uint8_to_normalized(xu).to(dev), lab.to(dev))["adv_output"]
    C.check("D(uint8) == D(normalised fp32)", o1, o2, 1e-6)
    C.finish()


@pytest.mark.parametrize("which", ["D", "G"])
@pytest.mark.parametrize("name", NAMES)
def test_bf16_vs_emulating_oracle(sg, name, which):
    bf16_vs_emulating_oracle(name, which)


class Taps:
    """cfg['taps'] of an oracle network function (oracle/restate.py _tap): keeps every block-boundary activation and, after backward, its gradient"""

    def __init__(self):
        self.act = {}

    def __call__(self, bi, act):
        act.retain_grad()
        self.act[bi] = act
        return act


class TeacherForcing:
    """net._sg_teacher (pytorch-studiogan_amd/ops.py block_boundary): records the output of every block and makes the next block read the tensor
    given for that boundary -- a fresh leaf, so the block's backward stops there"""

    def __init__(self, inputs):
        self.inputs, self.out, self.leaf = inputs, {}, {}

    def __call__(self, bi, act):
        self.out[bi] = act
        t = self.inputs.get(bi)
        if t is None:
            return act
        leaf = t.detach().clone().requires_grad_(True)
        self.leaf[bi] = leaf
        return leaf


class TapsReplace(TeacherForcing):
    """the same hook on the ORACLE side: the perturbed-weights run that measures the teacher-forced noise floor reads the unperturbed run's
    boundary activations"""


TEACHER_TOL = 1e-2     # relative-L2 bound of the teacher-forced block comparisons
FLOOR_EPS = 1e-5       # relative weight perturbation of the oracle's own noise-floor run (tools/bf16_noise_floor.py)
FLOOR_FACTOR = 1.5


def _floors_path(name):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".floors.json")


def _oracle_sha():
    """the floors are properties of the ORACLE alone (its own movement under a 1e-5 weight perturbation): a cached set is valid for the oracle source it was
    measured with"""
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return hashlib.sha256(open(os.path.join(here, "oracle", "restate.py"), "rb").read()).hexdigest()[:16]


def bf16_vs_emulating_oracle(name, which, report=None, batch=None, tg=None, floor=None, shared_objective=False, teacher=None, dev=None, floors_only=False, quad_emu=True,
                             teacher_base=None):
    """bf16 mode against the oracle run with `Bf16Emu` (oracle/restate.py): the same bf16 rounding at the same storage points
    (activations, activation gradients, weight images), fp32 everywhere else. Per operator the model is exact to 3e-5
    (conv / BN forward + backward) and 8e-3 (attention backward) -- tools/diag_bf16.py, measured on MI355X; over a whole
    network the residual is the ReLU-kink conditioning of the fp32 tests driven by those 3e-5 rounding-boundary disagreements
    instead of 1e-6 summation-order ones: measured 1-6 % relative-L2 on the width-8 fixtures, 4-10 % at the full DCGAN widths
    (against 12-35 % when the same bf16 run is compared with the fp32 oracle).

    batch: run the fixture's NETWORK on freshly seeded inputs of another batch size. tg: base gradient tolerance (relative-L2).
    floor (default: on for the full-width fixtures): MEASURED bound instead of a hand-set one. The emulating oracle is evaluated a second
    time with every weight perturbed by a relative 1e-5 (250 x below one bf16 ulp): however far IT moves -- per tensor -- is how far any two
    bf16 evaluations of this network that differ in a single rounding decision are apart, so each comparison is allowed
    max(base, 1.5 x that movement). For D the floor is 0.5-1 % and the base tolerance rules; for the generators it is 15-55 % at EVERY
    batch size (profiles/r03_bf16_batch_curve.txt): the weight gradient of the random linear functional used here is a random-walk sum
    over pixels (signal ~ sqrt(N)), and the units whose ReLU mask flips under rounding noise contribute sqrt(f N) to it, so the ratio does
    not fall with the batch. shared_objective=True uses ONE upstream-gradient image for all samples (partly coherent signal ~ N): there the
    floor does fall with the batch. Returns (whole-network gradient error, its floor).

    teacher (default: on for the backbones with block-boundary hooks, big_resnet / big_resnet_deep_legacy / resnet): the DISCRIMINATING bf16 check. The same
    oracle run keeps every block-boundary activation and its gradient (oracle/restate.py _tap); a second HIP pass feeds each block the oracle's
    input and upstream gradient (ops.block_boundary), so nothing compounds across blocks and no ReLU-mask flip of an earlier block reaches a later
    one: every block output, every block-input gradient and every weight gradient must agree to 1e-2 relative-L2 -- a 10-50 % error in one
    weight-gradient kernel, invisible under the whole-network floor of the generators, fails here."""
    dev = dev or torch.device("cuda:0")     # (the CPU when the package is bound to the interpreted library: tests/test_hipemu_net_cpu.py)
    fix, meta = load_golden(name)
    y = meta["yaml"]
    # quad_emu=False: the oracle keeps the REFERENCE graph's storage points for the convolutions next to a 2x resampling (one rounding per 3x3 filter
    # entry) instead of restating csrc/conv_q.h's second rounding of the summed filter: the comparison then bounds what that extra rounding costs
    # (ADVICE r4): pass teacher_base = the measured bound for it
    ocfg = dict(MG.oracle_cfg(y), emu=O.Bf16Emu, quad_emu=quad_emu)
    P, B = _split(sub(fix, which + "_init/"))
    _perturb(P, 7)
    if not floors_only:
        G, D = build_from_yaml(y, True, dev)
        net = D if which == "D" else G
        net.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)
        net.train()
        for p in net.parameters():
            p.grad = None
    C = Collector()
    full = name.endswith("w")
    if tg is None:
        tg = 0.15 if (meta.get("compact") or name in ("bigdeep32", "bigdeepsg32")) else 8e-2     # bigdeep32: 48 ReLU layers deep
        if full:
            tg = 0.3
    if floor is None:
        floor = full
    if batch is not None:
        gi = torch.Generator().manual_seed(1000 + batch)
        S_, nc_ = y["DATA"]["img_size"], y["DATA"]["num_classes"]
        fix = dict(fix)
        fix["in/real0"] = torch.randint(0, 256, (batch, 3, S_, S_), generator=gi).float() / 127.5 - 1.0
        fix["in/rl0"] = torch.randint(0, nc_, (batch,), generator=gi)
        fix["in/z0"] = torch.randn(batch, y["MODEL"].get("z_dim", 128), generator=gi)
        fix["in/fl0"] = torch.randint(0, nc_, (batch,), generator=gi)

    def oracle_run(leaves, taps=None):
        """-> dict of output tensors, input gradient (D) ; leaves carry .grad afterwards"""
        Bc = {k: v.clone() for k, v in B.items()}
        cfg_ = ocfg if taps is None else dict(ocfg, taps=taps)
        if which == "D":
            x, lab = fix["in/real0"].clone(), fix["in/rl0"]
            gadv = torch.tensor([0.3, -1.0, 0.7, 0.5, -0.2, 0.9, -0.6, 0.1]).repeat((x.shape[0] + 7) // 8)[:x.shape[0]]
            xo = x.clone().requires_grad_(True)
            adv_o, h_o = O.model_fns(cfg_)[1](xo, lab, leaves, Bc)
            (adv_o * gadv).sum().backward()
            return {"D adv": adv_o.detach(), "D h": h_o.detach(), "D dx": xo.grad}
        z, lab = fix["in/z0"], fix["in/fl0"]
        img_o = O.model_fns(cfg_)[0](z, lab, leaves, Bc, bn_mode="track")
        (img_o * gimg).sum().backward()
        return {"G img": img_o.detach()}

    if which == "G":
        S = y["DATA"]["img_size"]
        nb = fix["in/z0"].shape[0]
        gg = torch.Generator().manual_seed(11)
        gimg = torch.randn(1, 3, S, S, generator=gg).expand(nb, 3, S, S).contiguous() if shared_objective else torch.randn(nb, 3, S, S, generator=gg)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    if teacher is None:
        teacher = y["MODEL"]["backbone"] in ("big_resnet", "big_resnet_deep_legacy", "resnet")
    taps = Taps() if teacher else None
    ref = oracle_run(leaves, taps)
    fl = {}
    whole_floor = 0.0
    ckey = f"{which}|batch={batch}|shared={int(bool(shared_objective))}|teacher={int(bool(teacher))}" + ("" if quad_emu else "|quad_emu=0")
    cached = None
    if floor and not floors_only and os.path.exists(_floors_path(name)):
        ent = json.load(open(_floors_path(name))).get(ckey)
        if ent is not None and ent.get("oracle_sha16") == _oracle_sha():
            cached = ent
            fl, whole_floor = dict(ent["fl"]), float(ent["whole_floor"])
    if floor and cached is None:
        gp = torch.Generator().manual_seed(5)
        leaves2 = {k: (v * (1 + FLOOR_EPS * torch.randn(v.shape, generator=gp))).requires_grad_(True) for k, v in P.items()}
        ref2 = oracle_run(leaves2)
        gmax0 = max(float(v.grad.abs().max()) for v in leaves.values())
        num = den = 0.0
        for k in leaves:
            a, b = leaves2[k].grad.double(), leaves[k].grad.double()
            fk = (1.0 if k.endswith("sigma") else 1e-2) * gmax0 * (b.numel() ** 0.5)
            fl["grad " + k] = float((a - b).norm() / max(float(b.norm()), fk, 1e-30))
            num += float((a - b).pow(2).sum())
            den += float(b.pow(2).sum())
        whole_floor = (num / max(den, 1e-300)) ** 0.5
        for k in ref:
            fl[k] = float((ref2[k].double() - ref[k].double()).norm() / max(float(ref[k].double().norm()), 1e-30))

    tfl = {} if cached is None else dict(cached["tfl"])
    if teacher and taps.act and which == "G" and cached is None:
        # The generators' blocks hold two to four cBN + ReLU stages each: at the fixtures' batch of 2 a 4 x 4 map gives a channel 32 samples, the
        # batch-norm backward subtracts two means from the gradient, and a rounding-level disagreement flips ReLU units INSIDE the block. How far that
        # moves a teacher-forced quantity is measured like the whole-network floor: the oracle once more, weights perturbed by a relative 1e-5,
        # every block reading the UNPERTURBED run's input and upstream gradient. (The discriminators have no BN: their bound stays 1e-2 flat.)
        gp3 = torch.Generator().manual_seed(6)
        leaves3 = {k: (v * (1 + FLOOR_EPS * torch.randn(v.shape, generator=gp3))).requires_grad_(True) for k, v in P.items()}
        rep = TapsReplace({bi: a.detach() for bi, a in taps.act.items()})
        Bc3 = {k: v.clone() for k, v in B.items()}
        img3 = O.model_fns(dict(ocfg, taps=rep))[0](fix["in/z0"], fix["in/fl0"], leaves3, Bc3, bn_mode="track")
        order3 = sorted(rep.out)
        torch.autograd.backward([rep.out[b] for b in order3] + [img3], [taps.act[b].grad for b in order3] + [gimg])
        rl2 = lambda a, b, fk=0.0: float((a.double() - b.double()).norm() / max(float(b.double().norm()), fk, 1e-30))
        for b in order3:
            tfl[f"out {b}"] = rl2(rep.out[b].detach(), taps.act[b].detach())
            tfl[f"dx {b}"] = rl2(rep.leaf[b].grad, taps.act[b].grad)
        gmax3 = max(float(v.grad.abs().max()) for v in leaves.values())
        for k in leaves:
            tfl["grad " + k] = rl2(leaves3[k].grad, leaves[k].grad, (1.0 if k.endswith("sigma") else 1e-2) * gmax3 * (leaves[k].numel() ** 0.5))

    if floors_only:
        return {"key": ckey, "entry": {"oracle_sha16": _oracle_sha(), "fl": fl, "whole_floor": whole_floor, "tfl": tfl}}

    def tol(key, base):
        return max(base, FLOOR_FACTOR * fl.get(key, 0.0))

    if which == "D":
        xd = fix["in/real0"].to(dev).requires_grad_(True)
        lab = fix["in/rl0"]
        gadv = torch.tensor([0.3, -1.0, 0.7, 0.5, -0.2, 0.9, -0.6, 0.1]).repeat((xd.shape[0] + 7) // 8)[:xd.shape[0]]
        out = D(xd, lab.to(dev))
        (out["adv_output"] * gadv.to(dev)).sum().backward()
        (torch.cuda.synchronize() if dev.type == "cuda" else None)
        C.check("D adv", out["adv_output"], ref["D adv"], tol("D adv", 2e-2), l2=full)
        C.check("D h", out["h"], ref["D h"], tol("D h", 2e-2), l2=full)
        C.check("D dx", xd.grad, ref["D dx"], tol("D dx", tg), l2=True)
    else:
        img = G(fix["in/z0"].to(dev), fix["in/fl0"].to(dev))
        (img * gimg.to(dev)).sum().backward()
        (torch.cuda.synchronize() if dev.type == "cuda" else None)
        C.check("G img", img, ref["G img"], tol("G img", 2e-2), l2=full)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values())
    num = den = 0.0
    for k, p in net.named_parameters():
        # the attention gate is ONE scalar summing dy * conv(o) over every pixel: judged on the network's gradient scale
        C.check(which + " grad " + k, p.grad, leaves[k].grad, tol("grad " + k, tg), floor=(1.0 if k.endswith("sigma") else 1e-2) * gmax, l2=True)
        num += float((p.grad.detach().double().cpu() - leaves[k].grad.double()).pow(2).sum())
        den += float(leaves[k].grad.double().pow(2).sum())
    whole = (num / max(den, 1e-300)) ** 0.5        # relative-L2 of the WHOLE weight gradient of the network (all tensors concatenated)
    wt = max(tg, FLOOR_FACTOR * whole_floor)
    C.rows.append((which + " grad WHOLE-NETWORK", whole, wt))
    C.rows.append((which + " grad WHOLE-NETWORK oracle-own-floor", whole_floor, float("inf")))
    print(f"{which + ' grad WHOLE-NETWORK':52s} l2={whole:.3e} tol={wt:.1e} (oracle's own movement under a 1e-5 weight perturbation: {whole_floor:.3e}) {'ok' if whole <= wt else 'FAIL'}")
    def run_teacher(tag, factor):
        """one teacher-forced HIP pass; every comparison bounded by max(base, factor x the measured floor of that tensor)"""
        def ttol(key):
            base = 2e-2 if "conv1x1_" in key else (teacher_base or TEACHER_TOL)      # attention backward: 8e-3 per operator (tools/diag_bf16.py)
            return max(base, factor * tfl.get(key, 0.0))
        net.load_state_dict({**{k: v.to(dev) for k, v in P.items()}, **{k: v.to(dev) for k, v in B.items()}}, strict=True)   # u / v / BN state as the oracle saw them
        for p in net.parameters():
            p.grad = None
        to_dev = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
        tf = TeacherForcing({bi: to_dev(a) for bi, a in taps.act.items()})
        net.__dict__["_sg_teacher"] = tf
        from studiogan_amd import ops as _ops
        _ops.BLOCK_HOOKS[0] = True
        try:
            if which == "D":
                final = D(fix["in/real0"].to(dev), fix["in/rl0"].to(dev))["adv_output"]
                gfinal = gadv.to(dev)
            else:
                final = G(fix["in/z0"].to(dev), fix["in/fl0"].to(dev))
                gfinal = gimg.to(dev)
            order = sorted(b for b in tf.out if b in taps.act)
            roots = [tf.out[b] for b in order] + [final]
            grads = [to_dev(taps.act[b].grad) for b in order] + [gfinal.to(final.dtype) if final.dtype != gfinal.dtype else gfinal]
            torch.autograd.backward(roots, grads)
            (torch.cuda.synchronize() if dev.type == "cuda" else None)
        finally:
            _ops.BLOCK_HOOKS[0] = False
            net.__dict__.pop("_sg_teacher", None)
        back = lambda t: t.detach().float().permute(0, 3, 1, 2)
        pre = f"{which} teacher-forced{tag}"
        for b in order:
            C.check(f"{pre} block {b} out", back(tf.out[b]), taps.act[b].detach(), ttol(f"out {b}"), l2=True)
            if b in tf.leaf:
                gref = taps.act[b].grad
                C.check(f"{pre} dx into block {b + 1}", back(tf.leaf[b].grad), gref, ttol(f"dx {b}"), floor=1e-3 * float(gref.abs().max()), l2=True)
        C.check(f"{pre} final", final, ref["D adv"] if which == "D" else ref["G img"], TEACHER_TOL, l2=True)
        worst_tf = 0.0
        for k, p in net.named_parameters():
            e = C.check(f"{pre} grad " + k, p.grad, leaves[k].grad, ttol("grad " + k), floor=(1.0 if k.endswith("sigma") else 1e-2) * gmax, l2=True)
            worst_tf = max(worst_tf, e)
        print(f"{pre}: worst weight-gradient error {worst_tf:.3e}; worst measured floor {max(tfl.values()) if tfl else 0.0:.3e} (bound = max(1e-2, {factor} x floor) per tensor)")
        C.rows.append((pre + " WORST weight gradient", worst_tf, float("inf")))

    if teacher and taps.act:
        # ---- teacher-forced pass: every block on the oracle's own input and upstream gradient ------------------------------------------------
        # (The emulating oracle restates the quad kernels' filter rounding -- oracle/restate.py conv_pool_quad / conv_up_quad -- so the layers next
        # to a 2x resampling are modelled at the same storage points as every other layer. Before it did, session r4j / r4k: with SG_QUAD=0 every
        # generator tensor was within 1.5 x floor, with the quad kernels conv2d1's weight gradient sat at 2.6 x -- the filter-sum rounding flips more
        # ReLU units behind the cBN than a single-rounding model allows for.)
        run_teacher("", FLOOR_FACTOR)
        if tfl:
            C.rows.append((which + " teacher-forced WORST oracle-own-floor", max(tfl.values()), float("inf")))
    if report is not None:
        report.extend(C.rows)
        report.extend((which + " floor " + k, v, float("inf")) for k, v in fl.items())
    C.finish()
    return whole, whole_floor
