"""CPU oracle: a plain-torch (CPU, fp32/fp64) functional RESTATEMENT of the StudioGAN hot path.

TEST INFRASTRUCTURE ONLY. Nothing under studiogan_amd/ (the product path) imports this file; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker / timed CPU baseline.

Every function cites the reference lines it restates (paths relative to /root/reference/src). The restatement is
validated against the real reference (imported with stubs by oracle/ref_import.py in the authoring container) by
oracle/make_golden.py, which also writes the committed fixtures in tests/golden/.  Pin status: the reference has no
tests or golden vectors of its own; the pins are (i) outputs of the reference code itself run on CPU
(tests/golden/*.npz) and (ii) the parameter counts printed in the reference's training logs.

State convention: `P` maps reference parameter names to tensors, `B` maps buffer names to tensors (updated in place,
like the modules do). Names are the reference's state_dict keys.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------
# bf16 storage emulation. The HIP path's bf16 mode keeps activations, activation gradients and weight operand images in
# bf16 and everything else (accumulators, statistics, master weights, heads, optimizer) in fp32. `Bf16Emu` rounds at the
# same places so that bf16 parity can be asserted tightly (same ReLU masks up to rounding-boundary cases) instead of through
# the mask-flip noise bound against the fp32 oracle. With the default `IDENT` every hook returns its argument unchanged: the
# fp32 restatement is bit-for-bit what it was (make_golden.py re-checks that against the reference).
#   q  : value written to HBM by a kernel epilogue (round forward; the incoming gradient is a stored bf16 tensor: round backward)
#   qb : operand read by a kernel whose data gradient is written back in bf16 (identity forward, round backward)
#   qw : weight operand image (round forward, fp32 weight gradient passes through)
# ---------------------------------------------------------------------------------------------------------
def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _Ident:
    emulate = False
    q = qb = qw = staticmethod(lambda t: t)


class Bf16Emu:
    emulate = True
    q = staticmethod(_RoundBoth.apply)
    qb = staticmethod(_RoundBwd.apply)
    qw = staticmethod(_RoundFwd.apply)


IDENT = _Ident()


def emu_of(cfg):
    return cfg.get("emu") or IDENT


# ---------------------------------------------------------------------------------------------------------
# spectral norm -- torch.nn.utils.spectral_norm (installed torch/nn/utils/spectral_norm.py:92-114) as applied by
# utils/ops.py:195-224 with eps=1e-6, one power iteration per forward call while the module is in .train()
# ---------------------------------------------------------------------------------------------------------
def _l2n(v, eps):
    # F.normalize(v, dim=0, eps): v / max(||v||_2, eps)
    return v / torch.clamp(v.norm(), min=eps)


def weight_of(P, B, name, power_iterate=True, eps=1e-6, deconv=False):
    """Effective weight of layer `name`: W/sigma for spectral-norm layers (power iteration updates B in place)."""
    if name + ".weight_orig" not in P:
        return P[name + ".weight"]
    w = P[name + ".weight_orig"]
    u, v = B[name + ".weight_u"], B[name + ".weight_v"]
    # torch.nn.utils.spectral_norm picks dim=1 for ConvTranspose{1,2,3}d (spectral_norm.py:272-276): rows = out channels
    mat = w.permute(1, 0, 2, 3).reshape(w.shape[1], -1) if deconv else w.reshape(w.shape[0], -1)
    if power_iterate:
        with torch.no_grad():
            v_new = _l2n(mat.t().mv(u), eps)
            u_new = _l2n(mat.mv(v_new), eps)
            v.copy_(v_new)
            u.copy_(u_new)
    u_c, v_c = u.clone(), v.clone()
    sigma = torch.dot(u_c, mat.mv(v_c))
    return w / sigma


def conv(x, P, B, name, padding, sn_iter=True, E=IDENT, qb_in=True):
    """qb_in=False: the caller already placed the backward rounding (in front of a fused nearest-x2 upsample: the HIP data
    gradient sums the 2x2 block in fp32 and rounds once)."""
    w = weight_of(P, B, name, sn_iter)
    return F.conv2d(E.qb(x) if qb_in else x, E.qw(w), P.get(name + ".bias"), stride=1, padding=padding)


# ---- bf16 emulation of the quad kernels (csrc/conv_q.h): the 3x3 convolutions next to a 2x resampling ---------------------------------
# The HIP path computes avgpool2(conv3x3(x)) as ONE 4x4 / stride-2 convolution and conv3x3(up2(x)) as four 2x2 phase convolutions whose
# bf16 filter entries are the fp32 SUMS of the bf16 3x3 taps, rounded once more (sg_quad_pack). In exact arithmetic that is the reference's
# result (tests/test_quad_cpu.py); under Bf16Emu the second rounding is a storage point like any other and is restated here, for exactly the
# layers the kernels take (functional._quad_form: C % 32 == 0, Cout % 64 == 0 or % 96 == 0). The fp32 restatement (emu off) never comes here.
def _w(P, name):
    return P[name + ".weight_orig"] if name + ".weight_orig" in P else P[name + ".weight"]


def quad_eligible(w):
    return w.shape[1] % 32 == 0 and (w.shape[0] % 64 == 0 or w.shape[0] % 96 == 0) and tuple(w.shape[2:]) == (3, 3)


def conv_pool_quad(x, P, B, name, sn_iter, E):
    """E.q-free core of  avg_pool2d(conv3x3(x), 2)  as the kernel computes it: conv4x4 / stride 2 / pad 1 with w'[u][v] = 1/4 sum w[u-i][v-j]"""
    w = E.qw(weight_of(P, B, name, sn_iter))
    O_, I_ = w.shape[0], w.shape[1]
    w4 = F.conv2d(w.reshape(O_ * I_, 1, 3, 3), torch.full((1, 1, 2, 2), 0.25, dtype=w.dtype), padding=1).reshape(O_, I_, 4, 4)
    y = F.conv2d(E.qb(x), E.qw(w4), None, stride=2, padding=1)
    b = P.get(name + ".bias")
    return y if b is None else y + b.view(1, -1, 1, 1)


def conv_up_quad(x, P, B, name, sn_iter, E):
    """conv3x3(nearest_up2(x)) as the kernel computes it: output parity (a, b) = conv2x2(x; R_a w R_b^T), R_0 = [[1,0,0],[0,1,1]], R_1 = [[1,1,0],[0,0,1]]"""
    w = E.qw(weight_of(P, B, name, sn_iter))
    R = [torch.tensor([[1., 0., 0.], [0., 1., 1.]], dtype=w.dtype), torch.tensor([[1., 1., 0.], [0., 0., 1.]], dtype=w.dtype)]
    N, _, H, W = x.shape
    rows = []
    for a in (0, 1):
        cols = []
        for b in (0, 1):
            wq = E.qw(torch.einsum("tr,oirs,us->oitu", R[a], w, R[b]))
            cols.append(F.conv2d(F.pad(x, (1 - b, b, 1 - a, a)), wq))
        rows.append(torch.stack(cols, dim=-1).reshape(N, -1, H, 2 * W))          # interleave the two column parities
    y = torch.stack(rows, dim=-2).reshape(N, -1, 2 * H, 2 * W)                       # interleave the two row parities
    bia = P.get(name + ".bias")
    return y if bia is None else y + bia.view(1, -1, 1, 1)


def conv_strided(x, P, B, name, stride, padding, sn_iter=True, E=IDENT):
    return F.conv2d(E.qb(x), E.qw(weight_of(P, B, name, sn_iter)), P.get(name + ".bias"), stride=stride, padding=padding)


def deconv(x, P, B, name, stride, padding, sn_iter=True, E=IDENT):
    """nn.ConvTranspose2d (utils/ops.py:176-184); weight [Cin][Cout][kh][kw]. Spectral-norm variant (ops.py:207-216)
    normalises over dim 1."""
    return F.conv_transpose2d(E.qb(x), E.qw(weight_of(P, B, name, sn_iter, deconv=True)), P.get(name + ".bias"), stride=stride, padding=padding)


def linear(x, P, B, name, sn_iter=True):
    return F.linear(x, weight_of(P, B, name, sn_iter), P.get(name + ".bias"))


# ---------------------------------------------------------------------------------------------------------
# batch norm (utils/ops.py:227-228: eps 1e-4, momentum 0.1) and conditional batch norm (utils/ops.py:14-28)
# bn_mode: "track"   = .train() with running-stat updates (G-update, worker.py:513)
#          "untrack" = batch statistics, running stats untouched (D-update, worker.py:225, utils/misc.py:244-246)
#          "eval"    = running statistics
# ---------------------------------------------------------------------------------------------------------
def batch_norm(x, P, B, name, bn_mode, eps=1e-4, momentum=0.1):
    w, b = P.get(name + ".weight"), P.get(name + ".bias")
    if bn_mode == "untrack":
        return F.batch_norm(x, None, None, w, b, True, momentum, eps)
    rm, rv = B[name + ".running_mean"], B[name + ".running_var"]
    if bn_mode == "track":
        B[name + ".num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, w, b, True, momentum, eps)
    return F.batch_norm(x, rm, rv, w, b, False, momentum, eps)


def cond_batch_norm(x, y, P, B, name, bn_mode, sn_iter=True):
    gain = (1 + linear(y, P, B, name + ".gain", sn_iter)).view(y.size(0), -1, 1, 1)
    bias = linear(y, P, B, name + ".bias", sn_iter).view(y.size(0), -1, 1, 1)
    return batch_norm(x, P, B, name + ".bn", bn_mode) * gain + bias


# ---------------------------------------------------------------------------------------------------------
# self attention (utils/ops.py:83-103)
# ---------------------------------------------------------------------------------------------------------
def self_attention(x, P, B, name, sn_iter=True, E=IDENT):
    n, ch, h, w = x.shape
    theta = E.q(conv(x, P, B, name + ".conv1x1_theta", 0, sn_iter, E)).view(n, ch // 8, h * w)
    phi = F.max_pool2d(E.q(conv(x, P, B, name + ".conv1x1_phi", 0, sn_iter, E)), 2, 2).view(n, ch // 8, h * w // 4)
    attn = E.q(torch.softmax(E.qb(torch.bmm(theta.permute(0, 2, 1), phi)), dim=-1))
    g = F.max_pool2d(E.q(conv(x, P, B, name + ".conv1x1_g", 0, sn_iter, E)), 2, 2).view(n, ch // 2, h * w // 4)
    attn_g = E.q(torch.bmm(g, attn.permute(0, 2, 1))).view(n, ch // 2, h, w)
    attn_g = conv(attn_g, P, B, name + ".conv1x1_attn", 0, sn_iter, E)
    return E.q(x + P[name + ".sigma"] * attn_g)


# ---------------------------------------------------------------------------------------------------------
# BigGAN (models/big_resnet.py)
# ---------------------------------------------------------------------------------------------------------
def _tap(cfg, bi, act):
    """Block boundary `bi` of a network restatement (-1: in front of the first block): identity, unless the caller passed cfg['taps'] -- the
    teacher-forced block tests (tests/test_blocks_gpu.py) record the boundary activations and their gradients there."""
    t = cfg.get("taps")
    return act if t is None else t(bi, act)


def biggan_dims(img_size, ch):
    g_in = {32: [4, 4, 4], 64: [16, 8, 4, 2], 128: [16, 16, 8, 4, 2], 256: [16, 16, 8, 8, 4, 2]}[img_size]
    g_out = {32: [4, 4, 4], 64: [8, 4, 2, 1], 128: [16, 8, 4, 2, 1], 256: [16, 8, 8, 4, 2, 1]}[img_size]
    d_in = {32: [2, 2, 2], 64: [1, 2, 4, 8], 128: [1, 2, 4, 8, 16], 256: [1, 2, 4, 8, 8, 16]}[img_size]
    d_out = {32: [2, 2, 2, 2], 64: [1, 2, 4, 8, 16], 128: [1, 2, 4, 8, 16, 16], 256: [1, 2, 4, 8, 8, 16, 16]}[img_size]
    d_down = {32: [True, True, False, False], 64: [True] * 4 + [False], 128: [True] * 5 + [False], 256: [True] * 6 + [False]}[img_size]
    return [c * ch for c in g_in], [c * ch for c in g_out], [3] + [c * ch for c in d_in], [c * ch for c in d_out], d_down


def biggan_generator(z, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/big_resnet.py:122-158 (Generator.forward) + GenBlock.forward :28-42.
    cfg: dict(img_size, g_conv_dim, z_dim, attn_g_loc, apply_attn, g_cond_mtd)."""
    g_in, g_out, _, _, _ = biggan_dims(cfg["img_size"], cfg["g_conv_dim"])
    nb = len(g_in)
    chunk = cfg["z_dim"] // (nb + 1)
    zs = torch.split(z, chunk, 1)
    if cfg.get("g_cond_mtd", "cBN") != "W/O":
        shared = F.embedding(label, P["shared.weight"])
        affines = [torch.cat([shared, item], 1) for item in zs[1:]]
    else:
        affines = list(zs[1:])
    E = emu_of(cfg)
    act = _tap(cfg, -1, E.q(linear(zs[0], P, B, "linear0", sn_iter)).view(-1, g_in[0], 4, 4))
    bi = 0
    for index in range(nb):
        pre = f"blocks.{bi}.0"
        x0 = act
        x = E.q(torch.relu(cond_batch_norm(E.qb(act), affines[index], P, B, pre + ".bn1", bn_mode, sn_iter)))
        if E.emulate and cfg.get("quad_emu", True) and quad_eligible(_w(P, pre + ".conv2d1")):
            x = E.q(conv_up_quad(E.qb(x), P, B, pre + ".conv2d1", sn_iter, E))
        else:
            x = F.interpolate(E.qb(x), scale_factor=2, mode="nearest")
            x = E.q(conv(x, P, B, pre + ".conv2d1", 1, sn_iter, E, qb_in=False))
        x = E.q(torch.relu(cond_batch_norm(E.qb(x), affines[index], P, B, pre + ".bn2", bn_mode, sn_iter)))
        x = E.q(conv(x, P, B, pre + ".conv2d2", 1, sn_iter, E))
        x0 = conv(F.interpolate(E.qb(x0), scale_factor=2, mode="nearest"), P, B, pre + ".conv2d0", 0, sn_iter, E, qb_in=False)
        act = _tap(cfg, bi, E.q(x + x0))
        bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_g_loc"]:
            act = _tap(cfg, bi, self_attention(act, P, B, f"blocks.{bi}.0", sn_iter, E))
            bi += 1
    act = E.q(torch.relu(batch_norm(E.qb(act), P, B, "bn4", bn_mode)))
    return torch.tanh(E.q(conv(act, P, B, "conv2d5", 1, sn_iter, E)))


def biggan_discriminator(x, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/big_resnet.py:349-428 (Discriminator.forward), DiscOptBlock :177-192, DiscBlock :221-242.
    With apply_d_sn the blocks have no BN and nn.ReLU(inplace=True) (config.py:476) rewrites the block input, so the
    shortcut branch of DiscBlock ALSO sees relu(x) -- restated explicitly here (r = relu(x))."""
    _, _, d_in, d_out, d_down = biggan_dims(cfg["img_size"], cfg["d_conv_dim"])
    sn = cfg["apply_d_sn"]
    E = emu_of(cfg)
    h = E.q(x)
    bi = 0
    for index in range(len(d_in)):
        pre = f"blocks.{bi}.0"
        if index == 0:
            x0 = h
            y = E.q(conv(h, P, B, pre + ".conv2d1", 1, sn_iter, E))
            if not sn:
                y = E.q(torch.relu(batch_norm(E.qb(y), P, B, pre + ".bn1", bn_mode)))
                y = E.q(F.avg_pool2d(conv(y, P, B, pre + ".conv2d2", 1, sn_iter, E), 2))
            elif E.emulate and cfg.get("quad_emu", True) and quad_eligible(_w(P, pre + ".conv2d2")):
                y = E.q(conv_pool_quad(torch.relu(y), P, B, pre + ".conv2d2", sn_iter, E))
            else:
                y = E.q(F.avg_pool2d(conv(torch.relu(y), P, B, pre + ".conv2d2", 1, sn_iter, E), 2))
            x0 = E.q(F.avg_pool2d(E.qb(x0), 2))
            if not sn:
                x0 = E.q(batch_norm(E.qb(x0), P, B, pre + ".bn0", bn_mode))
            h = E.q(y + conv(x0, P, B, pre + ".conv2d0", 0, sn_iter, E))
        else:
            down, mismatch = d_down[index], d_in[index] != d_out[index]
            if sn:
                r = torch.relu(h)
                x0 = r
                y = r
            else:
                x0 = h
                y = E.q(torch.relu(batch_norm(E.qb(h), P, B, pre + ".bn1", bn_mode)))
            y = E.q(conv(y, P, B, pre + ".conv2d1", 1, sn_iter, E))
            if not sn:
                y = E.q(torch.relu(batch_norm(E.qb(y), P, B, pre + ".bn2", bn_mode)))
                y = conv(y, P, B, pre + ".conv2d2", 1, sn_iter, E)
            elif down and E.emulate and cfg.get("quad_emu", True) and quad_eligible(_w(P, pre + ".conv2d2")):
                y = conv_pool_quad(torch.relu(y), P, B, pre + ".conv2d2", sn_iter, E)
            else:
                y = conv(torch.relu(y), P, B, pre + ".conv2d2", 1, sn_iter, E)
                if down:
                    y = F.avg_pool2d(y, 2)
            if down and not sn:
                y = F.avg_pool2d(y, 2)
            y = E.q(y)
            if down or mismatch:
                if not sn:
                    x0 = E.q(batch_norm(E.qb(x0), P, B, pre + ".bn0", bn_mode))
                x0 = conv(x0, P, B, pre + ".conv2d0", 0, sn_iter, E)
                if down:
                    x0 = F.avg_pool2d(x0, 2)
            else:
                x0 = E.qb(x0)
            h = E.q(y + x0)
        h = _tap(cfg, bi, h)
        bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_d_loc"]:
            h = _tap(cfg, bi, self_attention(h, P, B, f"blocks.{bi}.0", sn_iter, E))
            bi += 1
    h = E.qb(h)
    h = torch.sum(torch.relu(h), dim=[2, 3])
    adv = torch.squeeze(linear(h, P, B, "linear1", sn_iter))
    if cfg.get("d_cond_mtd", "W/O") == "PD":
        emb = F.embedding(label, weight_of(P, B, "embedding", sn_iter))
        adv = adv + torch.sum(emb * h, 1)
    return adv, h


# ---------------------------------------------------------------------------------------------------------
# BigGAN-deep (models/big_resnet_deep_legacy.py): bottleneck blocks, channel-slice skip in G, channel-concat skip in D
# ---------------------------------------------------------------------------------------------------------
def biggan_deep_generator(z, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/big_resnet_deep_legacy.py:163-197 (Generator.forward), GenBlock.forward :52-78. One affine input
    [shared(label), z] for every block (no z chunks); g_depth blocks per stage, the last one upsamples."""
    g_in, g_out, _, _, _ = biggan_dims(cfg["img_size"], cfg["g_conv_dim"])
    E = emu_of(cfg)
    depth = cfg["g_depth"]
    if cfg.get("g_cond_mtd", "cBN") != "W/O":
        affine = torch.cat([F.embedding(label, P["shared.weight"]), z], 1)
    else:
        affine = z
    act = _tap(cfg, -1, E.q(linear(affine, P, B, "linear0", sn_iter)).view(-1, g_in[0], 4, 4))
    bi = 0
    for index in range(len(g_in)):
        for gi in range(depth):
            pre = f"blocks.{bi}.0"
            cin = g_in[index]
            cout = cin if gi == 0 else g_out[index]
            up = gi == depth - 1
            x0 = act[:, :cout] if cin != cout else act
            x = E.q(torch.relu(cond_batch_norm(E.qb(act), affine, P, B, pre + ".bn1", bn_mode, sn_iter)))
            x = E.q(conv(x, P, B, pre + ".conv2d1", 0, sn_iter, E))
            x = E.q(torch.relu(cond_batch_norm(E.qb(x), affine, P, B, pre + ".bn2", bn_mode, sn_iter)))
            if up and E.emulate and cfg.get("quad_emu", True) and quad_eligible(_w(P, pre + ".conv2d2")):
                x = E.q(conv_up_quad(E.qb(x), P, B, pre + ".conv2d2", sn_iter, E))
            else:
                if up:
                    x = F.interpolate(E.qb(x), scale_factor=2, mode="nearest")
                x = E.q(conv(x, P, B, pre + ".conv2d2", 1, sn_iter, E, qb_in=not up))
            x = E.q(torch.relu(cond_batch_norm(E.qb(x), affine, P, B, pre + ".bn3", bn_mode, sn_iter)))
            x = E.q(conv(x, P, B, pre + ".conv2d3", 1, sn_iter, E))
            x = E.q(torch.relu(cond_batch_norm(E.qb(x), affine, P, B, pre + ".bn4", bn_mode, sn_iter)))
            x = conv(x, P, B, pre + ".conv2d4", 0, sn_iter, E)
            if up:
                x0 = E.q(F.interpolate(E.qb(x0), scale_factor=2, mode="nearest"))
            act = _tap(cfg, bi, E.q(x + x0))
            bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_g_loc"]:
            act = _tap(cfg, bi, self_attention(act, P, B, f"blocks.{bi}.0", sn_iter, E))
            bi += 1
    act = E.q(torch.relu(batch_norm(E.qb(act), P, B, "bn4", bn_mode)))
    return torch.tanh(E.q(conv(act, P, B, "conv2d5", 1, sn_iter, E)))


def biggan_deep_dims(img_size, ch):
    d_in = {32: [4, 4, 4], 64: [1, 2, 4, 8], 128: [1, 2, 4, 8, 16], 256: [1, 2, 4, 8, 8, 16]}[img_size]
    d_out = {32: [4, 4, 4], 64: [2, 4, 8, 16], 128: [2, 4, 8, 16, 16], 256: [2, 4, 8, 8, 16, 16]}[img_size]
    d_down = {32: [True, True, False, False], 64: [True] * 4 + [False], 128: [True] * 5 + [False], 256: [True] * 6 + [False]}[img_size]
    return [c * ch for c in d_in], [c * ch for c in d_out], d_down


def biggan_deep_discriminator(x, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/big_resnet_deep_legacy.py:323-330,354-355 (Discriminator.forward), DiscBlock.forward :222-240.
    nn.ReLU(inplace=True) on the block input also rewrites the skip tensor (same storage): x0 = relu(x)."""
    d_in, d_out, d_down = biggan_deep_dims(cfg["img_size"], cfg["d_conv_dim"])
    E = emu_of(cfg)
    depth = cfg["d_depth"]
    h = _tap(cfg, -1, E.q(conv(E.q(x), P, B, "input_conv", 1, sn_iter, E)))
    bi = 0
    for index in range(len(d_in)):
        for di in range(depth):
            pre = f"blocks.{bi}.0"
            cin = d_in[index] if di == 0 else d_out[index]
            cout = d_out[index]
            down = d_down[index] and di == 0
            r = torch.relu(h)
            x0 = r
            y = E.q(conv(r, P, B, pre + ".conv2d1", 0, sn_iter, E))
            y = E.q(conv(torch.relu(y), P, B, pre + ".conv2d2", 1, sn_iter, E))
            y = E.q(conv(torch.relu(y), P, B, pre + ".conv2d3", 1, sn_iter, E))
            y = torch.relu(y)
            if down:
                y = F.avg_pool2d(y, 2)
            y = conv(y, P, B, pre + ".conv2d4", 0, sn_iter, E)
            if down:
                x0 = E.q(F.avg_pool2d(E.qb(x0), 2))
            if cin != cout:
                x0 = torch.cat([x0, E.q(conv(x0, P, B, pre + ".conv2d0", 0, sn_iter, E))], 1)
            h = _tap(cfg, bi, E.q(y + x0))
            bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_d_loc"]:
            h = _tap(cfg, bi, self_attention(h, P, B, f"blocks.{bi}.0", sn_iter, E))
            bi += 1
    h = torch.sum(torch.relu(E.qb(h)), dim=[2, 3])
    adv = torch.squeeze(linear(h, P, B, "linear1", sn_iter))
    if cfg.get("d_cond_mtd", "W/O") == "PD":
        adv = adv + torch.sum(F.embedding(label, weight_of(P, B, "embedding", sn_iter)) * h, 1)
    return adv, h


# ---------------------------------------------------------------------------------------------------------
# StudioGAN's BigGAN-deep variant (models/big_resnet_deep_studiogan.py): learned 1x1 skips, pool in front of the last ReLU of D
# ---------------------------------------------------------------------------------------------------------
def biggan_deep_sg_generator(z, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/big_resnet_deep_studiogan.py:143-176 (Generator.forward), GenBlock.forward :57-79: as the legacy generator, but the
    skip is conv2d0 (1x1) over the (upsampled) block input."""
    g_in, g_out, _, _, _ = biggan_dims(cfg["img_size"], cfg["g_conv_dim"])
    E = emu_of(cfg)
    depth = cfg["g_depth"]
    if cfg.get("g_cond_mtd", "cBN") != "W/O":
        affine = torch.cat([F.embedding(label, P["shared.weight"]), z], 1)
    else:
        affine = z
    act = E.q(linear(affine, P, B, "linear0", sn_iter)).view(-1, g_in[0], 4, 4)
    bi = 0
    for index in range(len(g_in)):
        for gi in range(depth):
            pre = f"blocks.{bi}.0"
            up = gi == depth - 1
            x0 = act
            x = E.q(torch.relu(cond_batch_norm(E.qb(act), affine, P, B, pre + ".bn1", bn_mode, sn_iter)))
            x = E.q(conv(x, P, B, pre + ".conv2d1", 0, sn_iter, E))
            x = E.q(torch.relu(cond_batch_norm(E.qb(x), affine, P, B, pre + ".bn2", bn_mode, sn_iter)))
            if up:
                x = F.interpolate(E.qb(x), scale_factor=2, mode="nearest")
            x = E.q(conv(x, P, B, pre + ".conv2d2", 1, sn_iter, E, qb_in=not up))
            x = E.q(torch.relu(cond_batch_norm(E.qb(x), affine, P, B, pre + ".bn3", bn_mode, sn_iter)))
            x = E.q(conv(x, P, B, pre + ".conv2d3", 1, sn_iter, E))
            x = E.q(torch.relu(cond_batch_norm(E.qb(x), affine, P, B, pre + ".bn4", bn_mode, sn_iter)))
            x = E.q(conv(x, P, B, pre + ".conv2d4", 0, sn_iter, E))
            if up:
                x0 = conv(F.interpolate(E.qb(x0), scale_factor=2, mode="nearest"), P, B, pre + ".conv2d0", 0, sn_iter, E, qb_in=False)
            else:
                x0 = conv(x0, P, B, pre + ".conv2d0", 0, sn_iter, E)
            act = E.q(x + x0)
            bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_g_loc"]:
            act = self_attention(act, P, B, f"blocks.{bi}.0", sn_iter, E)
            bi += 1
    act = E.q(torch.relu(batch_norm(E.qb(act), P, B, "bn4", bn_mode)))
    return torch.tanh(E.q(conv(act, P, B, "conv2d5", 1, sn_iter, E)))


def biggan_deep_sg_discriminator(x, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/big_resnet_deep_studiogan.py:323-330,354-355 (Discriminator.forward), DiscBlock.forward :232-250: the average
    pool precedes the last ReLU + conv1x1; the skip is conv2d0 over all output channels, pooled after the conv except in the
    first block (`optblock`: pool, then conv). nn.ReLU(inplace=True) on the block input rewrites the skip tensor: x0 = relu(x).
    32x32: the stem is d_conv_dim wide (:259)."""
    d_in, d_out, d_down = biggan_deep_dims(cfg["img_size"], cfg["d_conv_dim"])
    if cfg["img_size"] == 32:
        d_in = [cfg["d_conv_dim"]] + d_in[1:]
    E = emu_of(cfg)
    depth = cfg["d_depth"]
    h = E.q(conv(E.q(x), P, B, "input_conv", 1, sn_iter, E))
    bi = 0
    for index in range(len(d_in)):
        for di in range(depth):
            pre = f"blocks.{bi}.0"
            cin = d_in[index] if di == 0 else d_out[index]
            cout = d_out[index]
            down = d_down[index] and di == 0
            opt = index == 0 and di == 0
            r = torch.relu(h)
            x0 = r
            y = E.q(conv(r, P, B, pre + ".conv2d1", 0, sn_iter, E))
            y = E.q(conv(torch.relu(y), P, B, pre + ".conv2d2", 1, sn_iter, E))
            y = conv(torch.relu(y), P, B, pre + ".conv2d3", 1, sn_iter, E)
            if down:
                y = F.avg_pool2d(y, 2)
            y = E.q(y)
            y = conv(torch.relu(y), P, B, pre + ".conv2d4", 0, sn_iter, E)
            if opt:
                x0 = E.q(conv(E.q(F.avg_pool2d(E.qb(x0), 2)), P, B, pre + ".conv2d0", 0, sn_iter, E))
            elif down or cin != cout:
                x0 = conv(x0, P, B, pre + ".conv2d0", 0, sn_iter, E)
                if down:
                    x0 = F.avg_pool2d(x0, 2)
                x0 = E.q(x0)
            h = E.q(y + x0)
            bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_d_loc"]:
            h = self_attention(h, P, B, f"blocks.{bi}.0", sn_iter, E)
            bi += 1
    h = torch.sum(torch.relu(E.qb(h)), dim=[2, 3])
    adv = torch.squeeze(linear(h, P, B, "linear1", sn_iter))
    if cfg.get("d_cond_mtd", "W/O") == "PD":
        adv = adv + torch.sum(F.embedding(label, weight_of(P, B, "embedding", sn_iter)) * h, 1)
    return adv, h


# ---------------------------------------------------------------------------------------------------------
# SNGAN-style ResNet generator (models/resnet.py:15-158); its discriminator is line-for-line models/big_resnet.py's
# ---------------------------------------------------------------------------------------------------------
def resnet_generator(z, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/resnet.py:126-158 (Generator.forward), GenBlock.forward :36-60. Conditioning: cBN on the ONE-HOT label
    (resnet.py:128-129,140-141) -- no shared embedding, no z chunks; unconditional: plain BN (resnet.py:21-23)."""
    g_in, g_out, _, _, _ = biggan_dims(cfg["img_size"], cfg["g_conv_dim"])
    cond = cfg.get("g_cond_mtd", "W/O") != "W/O"
    aff = F.one_hot(label, num_classes=cfg["num_classes"]).to(torch.float32) if cond else None

    E = emu_of(cfg)

    def bn(x, name):
        x = E.qb(x)
        return cond_batch_norm(x, aff, P, B, name, bn_mode, sn_iter) if cond else batch_norm(x, P, B, name, bn_mode)

    act = _tap(cfg, -1, E.q(linear(z, P, B, "linear0", sn_iter)).view(-1, g_in[0], 4, 4))
    bi = 0
    for index in range(len(g_in)):
        pre = f"blocks.{bi}.0"
        x0 = act
        x = E.q(torch.relu(bn(act, pre + ".bn1")))
        if E.emulate and cfg.get("quad_emu", True) and quad_eligible(_w(P, pre + ".conv2d1")):
            x = E.q(conv_up_quad(E.qb(x), P, B, pre + ".conv2d1", sn_iter, E))      # (the HIP path's storage points: csrc/conv_q.h rounds the phase filters once more)
        else:
            x = E.q(conv(F.interpolate(E.qb(x), scale_factor=2, mode="nearest"), P, B, pre + ".conv2d1", 1, sn_iter, E, qb_in=False))
        x = E.q(conv(E.q(torch.relu(bn(x, pre + ".bn2"))), P, B, pre + ".conv2d2", 1, sn_iter, E))
        x0 = conv(F.interpolate(E.qb(x0), scale_factor=2, mode="nearest"), P, B, pre + ".conv2d0", 0, sn_iter, E, qb_in=False)
        act = _tap(cfg, bi, E.q(x + x0))
        bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_g_loc"]:
            act = _tap(cfg, bi, self_attention(act, P, B, f"blocks.{bi}.0", sn_iter, E))
            bi += 1
    act = E.q(torch.relu(batch_norm(E.qb(act), P, B, "bn4", bn_mode)))
    return torch.tanh(E.q(conv(act, P, B, "conv2d5", 1, sn_iter, E)))


# ---------------------------------------------------------------------------------------------------------
# DCGAN (models/deep_conv.py): fixed widths 512-256-128-64 / 64-128-256-512, 32x32 images
# ---------------------------------------------------------------------------------------------------------
def dcgan_generator(z, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/deep_conv.py:90-121 (Generator.forward), GenBlock.forward :32-39."""
    cond = cfg.get("g_cond_mtd", "W/O") != "W/O"
    aff = F.one_hot(label, num_classes=cfg["num_classes"]).to(torch.float32) if cond else None
    E = emu_of(cfg)
    act = E.q(linear(z, P, B, "linear0", sn_iter)).view(-1, 512, 4, 4)
    bi = 0
    for index in range(3):
        pre = f"blocks.{bi}.0"
        x = E.qb(E.q(deconv(act, P, B, pre + ".deconv0", 2, 1, sn_iter, E)))
        x = cond_batch_norm(x, aff, P, B, pre + ".bn0", bn_mode, sn_iter) if cond else batch_norm(x, P, B, pre + ".bn0", bn_mode)
        act = E.q(torch.relu(x))
        bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_g_loc"]:
            act = self_attention(act, P, B, f"blocks.{bi}.0", sn_iter, E)
            bi += 1
    return torch.tanh(E.q(conv(act, P, B, "conv4", 1, sn_iter, E)))


def dcgan_discriminator(x, label, P, B, cfg, bn_mode="track", sn_iter=True):
    """models/deep_conv.py:232-247,270-271 (Discriminator.forward), DiscBlock.forward :139-149."""
    sn = cfg["apply_d_sn"]
    E = emu_of(cfg)
    h = E.q(x)
    bi = 0
    for index in range(3):
        pre = f"blocks.{bi}.0"
        h = E.q(conv(h, P, B, pre + ".conv0", 1, sn_iter, E))
        if not sn:
            h = E.q(torch.relu(batch_norm(E.qb(h), P, B, pre + ".bn0", bn_mode)))
            h = E.q(conv_strided(h, P, B, pre + ".conv1", 2, 1, sn_iter, E))
            h = E.q(torch.relu(batch_norm(E.qb(h), P, B, pre + ".bn1", bn_mode)))
        else:
            h = E.q(conv_strided(torch.relu(h), P, B, pre + ".conv1", 2, 1, sn_iter, E))
            h = torch.relu(h)
        bi += 1
        if cfg.get("apply_attn", False) and (index + 1) in cfg["attn_d_loc"]:
            h = self_attention(h, P, B, f"blocks.{bi}.0", sn_iter, E)
            bi += 1
    h = E.q(conv(h, P, B, "conv1", 1, sn_iter, E))
    if not sn:
        h = E.q(batch_norm(E.qb(h), P, B, "bn1", bn_mode))
    h = torch.sum(torch.relu(E.qb(h)), dim=[2, 3])
    adv = torch.squeeze(linear(h, P, B, "linear1", sn_iter))
    if cfg.get("d_cond_mtd", "W/O") == "PD":
        adv = adv + torch.sum(F.embedding(label, weight_of(P, B, "embedding", sn_iter)) * h, 1)
    return adv, h


# ---------------------------------------------------------------------------------------------------------
# losses (utils/losses.py:197-239) / gradient penalty (utils/losses.py:268-275,301-316)
# ---------------------------------------------------------------------------------------------------------
def feature_matching(real_h, fake_h):
    """utils/losses.py:254-259"""
    return torch.mean(torch.abs(torch.mean(fake_h, 0) - torch.mean(real_h, 0)))


def d_loss(kind, real, fake):
    if kind == "hinge":
        return torch.mean(F.relu(1. - real)) + torch.mean(F.relu(1. + fake))
    if kind == "wasserstein":
        return torch.mean(fake - real)
    if kind == "vanilla":
        return torch.mean(F.softplus(-real)) + torch.mean(F.softplus(fake))
    if kind == "logistic":           # utils/losses.py:207-209
        return (F.softplus(-real) + F.softplus(fake)).mean()
    if kind == "least_square":       # utils/losses.py:216-218
        return (0.5 * (real - torch.ones_like(real)) ** 2 + 0.5 * fake ** 2).mean()
    raise ValueError(kind)


def g_loss(kind, fake):
    if kind in ("hinge", "wasserstein"):
        return -torch.mean(fake)
    if kind in ("vanilla", "logistic"):
        return torch.mean(F.softplus(-fake))
    if kind == "least_square":       # utils/losses.py:221-223
        return (0.5 * (fake - torch.ones_like(fake)) ** 2).mean()
    raise ValueError(kind)


def grad_penalty(dis_fn, real, real_labels, fake, P, B, alpha):
    """utils/losses.py:301-316 with alpha [B,1] given (the reference draws it with torch.rand(batch_size, 1) on the host
    RNG, :303): penalty = mean_b (||d sum(D(x_b)) / d x_b||_2 - 1)^2 at x = alpha*real + (1-alpha)*fake."""
    a = alpha.view(-1, 1, 1, 1)
    x = (a * real + (1 - a) * fake).detach().requires_grad_(True)
    adv, _ = dis_fn(x, real_labels, P, B)
    g = torch.autograd.grad(outputs=adv, inputs=x, grad_outputs=torch.ones_like(adv), create_graph=True, retain_graph=True, only_inputs=True)[0]
    g = g.view(g.size(0), -1)
    return ((g.norm(2, dim=1) - 1) ** 2).mean() + x[:, 0, 0, 0].mean() * 0


def maxgrad_penalty(dis_fn, real, real_labels, fake, P, B, alpha):
    """utils/losses.py:338-352: the interpolates of grad_penalty, penalty = max_b ||grad_b||^2."""
    a = alpha.view(-1, 1, 1, 1)
    x = (a * real + (1 - a) * fake).detach().requires_grad_(True)
    adv, _ = dis_fn(x, real_labels, P, B)
    g = torch.autograd.grad(outputs=adv, inputs=x, grad_outputs=torch.ones_like(adv), create_graph=True, retain_graph=True, only_inputs=True)[0]
    g = g.view(g.size(0), -1)
    return torch.max(g.norm(2, dim=1) ** 2) + x[:, 0, 0, 0].mean() * 0


def dra_penalty(dis_fn, real, real_labels, P, B, alpha, noise):
    """utils/losses.py:319-335 with the host draws given (alpha = torch.rand(B,1,1,1), noise = torch.rand(real.size()), :321,325):
    the gradient-penalty functional at real + alpha * 0.5 * real.std() * noise."""
    x = (real + alpha * (0.5 * real.std() * noise)).detach().requires_grad_(True)
    adv, _ = dis_fn(x, real_labels, P, B)
    g = torch.autograd.grad(outputs=adv, inputs=x, grad_outputs=torch.ones_like(adv), create_graph=True, retain_graph=True, only_inputs=True)[0]
    g = g.view(g.size(0), -1)
    return ((g.norm(2, dim=1) - 1) ** 2).mean() + x[:, 0, 0, 0].mean() * 0


def lecam_reg(d_logit_real, d_logit_fake, ema_d_real, ema_d_fake):
    """utils/losses.py:262-265 (ema.D_real / ema.D_fake: utils/ops.py:106-133)."""
    return torch.mean(F.relu(d_logit_real - ema_d_fake).pow(2)) + torch.mean(F.relu(ema_d_real - d_logit_fake).pow(2))


def uint8_to_normalized(x_u8_nhwc, flip=None):
    """data_util.py:92-94,141: ToTensor (HWC uint8 -> CHW float / 255) + Normalize(0.5, 0.5) (+ horizontal flip) -> fp32 NCHW."""
    x = x_u8_nhwc.permute(0, 3, 1, 2).to(torch.float32).div(255.0)
    x = (x - 0.5) / 0.5
    if flip is not None:
        x = torch.where(flip.view(-1, 1, 1, 1).bool(), x.flip(3), x)
    return x


def r1_reg(dis_fn, real, real_labels, P, B):
    """utils/losses.py:355-361 on the real batch (src/worker.py:260-261,410-412): 0.5 mean_b ||d sum(D(x)) / d x_b||^2.
    Returns (penalty, adv) -- the reference takes the gradient through the SAME forward that feeds the adversarial loss."""
    x = real.detach().requires_grad_(True)
    adv, _ = dis_fn(x, real_labels, P, B)
    g = torch.autograd.grad(outputs=adv.sum(), inputs=x, grad_outputs=torch.ones([]), create_graph=True, retain_graph=True, only_inputs=True)[0]
    r1 = 0.5 * g.pow(2).contiguous().view(real.shape[0], -1).sum(1).mean(0) + x[:, 0, 0, 0].mean() * 0
    return r1, adv


# ---------------------------------------------------------------------------------------------------------
# optimizer / EMA (config.py:541-563 torch.optim.Adam eps=1e-6; utils/ema.py:27-40)
# ---------------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1, beta2, eps=1e-6):
    """In-place torch.optim.Adam single-tensor update (installed torch/optim/adam.py _single_tensor_adam)."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_update(src_P, src_B, ema_P, ema_B, step, decay=0.9999, start_iter=0):
    d = 0.0 if (0 <= step < start_iter) else decay
    with torch.no_grad():
        for k in ema_P:
            ema_P[k].copy_(src_P[k].lerp(ema_P[k], d))
        for k in ema_B:
            if "num_batches_tracked" in k:
                ema_B[k].copy_(src_B[k])
            else:
                ema_B[k].copy_(src_B[k].lerp(ema_B[k], d))


# ---------------------------------------------------------------------------------------------------------
# one training step, re-enacting worker.py:213-497 (train_discriminator) and :502-681 (train_generator) for the
# conv-GAN family without augmentation: returns losses; updates P/B/optimizer state in place
# ---------------------------------------------------------------------------------------------------------
class AdamState:
    def __init__(self, P, lr, beta1, beta2, eps=1e-6):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.t = 0

    def step(self, P, grads):
        self.t += 1
        with torch.no_grad():
            for k, p in P.items():
                if grads.get(k) is None:
                    continue
                adam_step(p, grads[k], self.m[k], self.v[k], self.t, self.lr, self.b1, self.b2, self.eps)


def _leaves(P):
    return {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}


def d_update(gen_fn, dis_fn, GP, GB, DP, DB, d_opt, real, real_labels, z, fake_labels, loss_kind="hinge", record=False, gp_lambda=None,
             gp_alpha=None):
    """One discriminator update on pre-drawn (z, fake_labels, real) micro-batches (lists of length acml).
    G runs in train mode without BN-stat tracking and without a graph (worker.py:216-225)."""
    acml = len(z)
    leaves = _leaves(DP)
    out = {"loss": 0.0}
    for i in range(acml):
        with torch.no_grad():
            fake = gen_fn(z[i], fake_labels[i], GP, GB, bn_mode="untrack")
        adv_r, _ = dis_fn(real[i], real_labels[i], leaves, DB)
        adv_f, _ = dis_fn(fake, fake_labels[i], leaves, DB)
        loss = d_loss(loss_kind, adv_r, adv_f)
        if gp_lambda is not None:   # worker.py:369-375
            gp = grad_penalty(dis_fn, real[i], real_labels[i], fake, leaves, DB, gp_alpha[i])
            loss = loss + gp_lambda * gp
            out["gp"] = gp.detach()
        loss = loss / acml
        loss.backward()
        out["loss"] += float(loss.detach())
        if record and i == 0:
            out["fake"], out["adv_r"], out["adv_f"] = fake.detach(), adv_r.detach(), adv_f.detach()
    grads = {k: v.grad for k, v in leaves.items()}
    out["grads"] = grads
    d_opt.step(DP, grads)
    return out


def g_update(gen_fn, dis_fn, GP, GB, DP, DB, g_opt, z, fake_labels, loss_kind="hinge", record=False):
    """One generator update (worker.py:502-681): G tracks BN stats; D weights frozen but its SN still iterates."""
    acml = len(z)
    leaves = _leaves(GP)
    out = {"loss": 0.0}
    for i in range(acml):
        fake = gen_fn(z[i], fake_labels[i], leaves, GB, bn_mode="track")
        adv_f, _ = dis_fn(fake, fake_labels[i], DP, DB)
        loss = g_loss(loss_kind, adv_f) / acml
        loss.backward()
        out["loss"] += float(loss.detach())
        if record and i == 0:
            out["fake"], out["adv_f"] = fake.detach(), adv_f.detach()
    grads = {k: v.grad for k, v in leaves.items()}
    out["grads"] = grads
    g_opt.step(GP, grads)
    return out


def model_fns(cfg):
    """(generator_fn, discriminator_fn) closures for a configuration dict (see oracle/make_golden.py:oracle_cfg)."""
    bb = cfg.get("backbone", "big_resnet")
    if bb == "big_resnet":
        def gen_fn(z, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_generator(z, y, P, B, cfg, bn_mode, sn_iter)

        def dis_fn(x, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_discriminator(x, y, P, B, cfg, bn_mode, sn_iter)
        return gen_fn, dis_fn
    if bb == "big_resnet_deep_legacy":
        def gen_fn(z, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_deep_generator(z, y, P, B, cfg, bn_mode, sn_iter)

        def dis_fn(x, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_deep_discriminator(x, y, P, B, cfg, bn_mode, sn_iter)
        return gen_fn, dis_fn
    if bb == "big_resnet_deep_studiogan":
        def gen_fn(z, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_deep_sg_generator(z, y, P, B, cfg, bn_mode, sn_iter)

        def dis_fn(x, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_deep_sg_discriminator(x, y, P, B, cfg, bn_mode, sn_iter)
        return gen_fn, dis_fn
    if bb == "resnet":
        def gen_fn(z, y, P, B, bn_mode="track", sn_iter=True):
            return resnet_generator(z, y, P, B, cfg, bn_mode, sn_iter)

        def dis_fn(x, y, P, B, bn_mode="track", sn_iter=True):
            return biggan_discriminator(x, y, P, B, cfg, bn_mode, sn_iter)
        return gen_fn, dis_fn
    if bb == "deep_conv":
        def gen_fn(z, y, P, B, bn_mode="track", sn_iter=True):
            return dcgan_generator(z, y, P, B, cfg, bn_mode, sn_iter)

        def dis_fn(x, y, P, B, bn_mode="track", sn_iter=True):
            return dcgan_discriminator(x, y, P, B, cfg, bn_mode, sn_iter)
        return gen_fn, dis_fn
    raise NotImplementedError(bb)


# ---------------------------------------------------------------------------------------------------------
# class-conditioning heads and losses (models/big_resnet.py:307-333,358-413; utils/losses.py:40-165,242-252)
# ---------------------------------------------------------------------------------------------------------
def d_heads(adv, h, label, P, B, cfg, adc_fake=False, sn_iter=True):
    """Everything of Discriminator.forward after `adv_output = squeeze(linear1(h))` (big_resnet.py:363-413) except the projection term
    (already in the trunk functions). Returns the dictionary the reference returns (the entries the losses read)."""
    mtd, aux, nrm = cfg.get("d_cond_mtd", "W/O"), cfg.get("aux_cls_type", "W/O"), cfg.get("normalize_d_embed", False)
    out = {"h": h, "adv_output": adv, "embed": None, "proxy": None, "cls_output": None, "label": label,
           "mi_embed": None, "mi_proxy": None, "mi_cls_output": None}
    if aux == "ADC":
        label = label * 2 + 1 if adc_fake else label * 2
        out["label"] = label
    if mtd == "AC":
        if nrm:
            h = F.normalize(h, dim=1)          # (the weight normalisation loop of the reference is a no-op: it rebinds a local name)
            out["h"] = h
        out["cls_output"] = linear(h, P, B, "linear2", sn_iter)
    elif mtd in ("2C", "D2DCE"):
        embed = linear(h, P, B, "linear2", sn_iter)
        proxy = F.embedding(label, weight_of(P, B, "embedding", sn_iter))
        if nrm:
            embed, proxy = F.normalize(embed, dim=1), F.normalize(proxy, dim=1)
        out["embed"], out["proxy"] = embed, proxy
    elif mtd == "MD":
        out["adv_output"] = adv[torch.arange(label.shape[0]), label]
    if aux == "TAC":
        if mtd == "AC":
            out["mi_cls_output"] = linear(h, P, B, "linear_mi", sn_iter)
        elif mtd in ("2C", "D2DCE"):
            mi_e = linear(h, P, B, "linear_mi", sn_iter)
            mi_p = F.embedding(label, weight_of(P, B, "embedding_mi", sn_iter))
            if nrm:
                mi_e, mi_p = F.normalize(mi_e, dim=1), F.normalize(mi_p, dim=1)
            out["mi_embed"], out["mi_proxy"] = mi_e, mi_p
    return out


def _cos_matrix(x, y):
    return F.cosine_similarity(x.unsqueeze(1), y.unsqueeze(0), dim=-1)


def _off_diag(M):
    n = M.shape[0]
    return M[~torch.eye(n, dtype=torch.bool)].view(n, n - 1)


def cond_loss(mtd, out, temperature=1.0, m_p=1.0):
    """utils/losses.py:40-47 (AC: cross entropy), :50-97 (2C), :100-165 (D2DCE) on a head dictionary."""
    label = out["label"]
    if mtd == "AC":
        return F.cross_entropy(out["cls_output"], label)
    embed, proxy = out["embed"], out["proxy"]
    same = (label.view(-1, 1) == label.view(1, -1))
    if mtd == "2C":
        sim = torch.exp(_off_diag(_cos_matrix(embed, embed)) / temperature)
        pos = _off_diag(same.long()) * sim
        e2p = torch.exp(F.cosine_similarity(embed, proxy, dim=-1) / temperature)
        num = e2p + pos.sum(dim=1)
        den = torch.cat([e2p.unsqueeze(1), sim], dim=1).sum(dim=1)
        return -torch.log(num / den).mean()
    if mtd == "D2DCE":
        sim = _off_diag((_cos_matrix(embed, embed) + m_p - 1) / temperature)
        mx, _ = torch.max(sim, dim=1, keepdim=True)
        sim = F.relu(sim) - mx.detach()
        s2p = F.cosine_similarity(embed, proxy, dim=-1)
        improved = _off_diag((~same).long()) * torch.exp(sim)
        pos_attr = F.relu((m_p - s2p) / temperature)
        neg_repul = torch.log(torch.exp(-pos_attr) + improved.sum(dim=1))
        return (pos_attr + neg_repul).mean()
    raise NotImplementedError(mtd)


def crammer_singer(adv, label):
    """utils/losses.py:242-252."""
    n = adv.shape[1] - 1
    mask = torch.ones_like(adv)
    mask.scatter_(1, label.unsqueeze(-1), 0)
    wrongs = torch.masked_select(adv, mask.bool()).reshape(adv.shape[0], n)
    max_wrong = wrongs.max(1)[0].unsqueeze(-1)
    target = adv.gather(1, label.unsqueeze(-1))
    return torch.mean(F.relu(1 + max_wrong - target))


def d_side_loss(dis_fn, P, B, cfg, real, real_labels, fake, fake_labels, loss_kind, hp):
    """The discriminator-side loss of one micro-batch with class conditioning (src/worker.py:281-317): adversarial loss (or the multi-hinge
    pair) + cond_lambda * conditioning loss on the real half (+ the TAC / ADC term on the fake half). hp: cond_lambda, temperature, m_p,
    tac_dis_lambda. Returns (loss, real head dictionary, fake head dictionary)."""
    mtd, aux = cfg.get("d_cond_mtd", "W/O"), cfg.get("aux_cls_type", "W/O")
    adv_r, h_r = dis_fn(real, real_labels, P, B)
    rd = d_heads(adv_r, h_r, real_labels, P, B, cfg, False)
    adv_f, h_f = dis_fn(fake, fake_labels, P, B)
    fd = d_heads(adv_f, h_f, fake_labels, P, B, cfg, aux == "ADC")
    if loss_kind == "MH":
        lossy = torch.full((fake.shape[0],), cfg["num_classes"], dtype=torch.long)
        loss = crammer_singer(rd["adv_output"], rd["label"]) + crammer_singer(fd["adv_output"], lossy)
    else:
        loss = d_loss(loss_kind, rd["adv_output"], fd["adv_output"])
    if mtd in ("AC", "2C", "D2DCE"):
        loss = loss + hp["cond_lambda"] * cond_loss(mtd, rd, hp["temperature"], hp["m_p"])
        if aux == "TAC":
            loss = loss + hp["tac_dis_lambda"] * cond_loss(mtd, fd, hp["temperature"], hp["m_p"])     # (src/worker.py:311: the twin loss reads cls_output / embed / proxy)
        elif aux == "ADC":
            loss = loss + hp["cond_lambda"] * cond_loss(mtd, fd, hp["temperature"], hp["m_p"])
    return loss, rd, fd


# ---------------------------------------------------------------------------------------------------------
# DiffAugment + consistency regularisers around the discriminator (worker.py:236-365,541-603)
# ---------------------------------------------------------------------------------------------------------
def _consistency(a, b, mtd):
    """torch.nn.MSELoss between the two views' logits (+ class logits for AC, embeddings for 2C / D2D-CE): worker.py:329-335,344-353,358-364"""
    loss = ((a["adv_output"] - b["adv_output"]) ** 2).mean()
    if mtd == "AC":
        loss = loss + ((a["cls_output"] - b["cls_output"]) ** 2).mean()
    elif mtd in ("2C", "D2DCE"):
        loss = loss + ((a["embed"] - b["embed"]) ** 2).mean()
    return loss


def d_consistency_loss(gen_fn, dis_fn, GP, GB, P, B, cfg, real, real_labels, z, fake_labels, loss_kind, hp, draws, z_eps=None):
    """The discriminator-side loss of one micro-batch with DiffAugment in front of the discriminator and the consistency regularisers behind it
    (src/worker.py:236-365), in the reference's order of generator / discriminator forwards (the spectral-norm vectors advance with each):
    G(z), [G(z + eps)] (utils/sample.py:162-176), D(series(real)), D(series(fake)), [CR: D(parallel(real))], [bCR: D(parallel(real)), D(parallel(fake))],
    [zCR: D(G(z + eps))]. hp: diffaug_policy (or None), cr_lambda / (real_lambda, fake_lambda) / d_lambda (each None = off). draws: the random
    numbers of each augmentation call in consumption order ("series_real", "series_fake": oracle/aug_ref.draw_diffaug lists; "prl_real", "prl_fake":
    draw_cr triples). Returns (loss, fake images)."""
    from . import aug_ref as AR
    mtd = cfg.get("d_cond_mtd", "W/O")
    with torch.no_grad():
        fake = gen_fn(z, fake_labels, GP, GB, bn_mode="untrack")
        fake_eps = gen_fn(z_eps, fake_labels, GP, GB, bn_mode="untrack") if z_eps is not None else None
    series = (lambda x, d: AR.diffaug(x, hp["diffaug_policy"], d)) if hp.get("diffaug_policy") else (lambda x, d: x)
    if hp.get("apa_p") is not None:          # worker.py:273-274
        real = AR.apa(real, fake, draws["apa"][0], hp["apa_p"])

    def heads(x, lab):
        adv, h = dis_fn(x, lab, P, B)
        return d_heads(adv, h, lab, P, B, cfg, False)
    rd = heads(series(real, draws.get("series_real")), real_labels)
    fd = heads(series(fake, draws.get("series_fake")), fake_labels)
    loss = d_loss(loss_kind, rd["adv_output"], fd["adv_output"])
    if hp.get("cr_lambda") is not None:
        loss = loss + hp["cr_lambda"] * _consistency(rd, heads(AR.cr_aug(real, *draws["prl_real"]), real_labels), mtd)
    if hp.get("bcr_lambdas") is not None:
        real_prl, fake_prl = AR.cr_aug(real, *draws["prl_real"]), AR.cr_aug(fake, *draws["prl_fake"])
        rp = heads(real_prl, real_labels)
        fp = heads(fake_prl, fake_labels)
        loss = loss + hp["bcr_lambdas"][0] * _consistency(rd, rp, mtd) + hp["bcr_lambdas"][1] * _consistency(fd, fp, mtd)
    if hp.get("d_lambda") is not None:
        loss = loss + hp["d_lambda"] * _consistency(fd, heads(fake_eps, fake_labels), mtd)
    return loss, fake


def g_consistency_loss(gen_fn, dis_fn, GP, GB, DP, DB, cfg, z, fake_labels, loss_kind, hp, draws, z_eps=None):
    """The generator-side loss (src/worker.py:520-603): G(z), [G(z + eps)], D(series(fake)), adversarial loss - g_lambda * MSE(G(z), G(z + eps))."""
    from . import aug_ref as AR
    fake = gen_fn(z, fake_labels, GP, GB, bn_mode="track")
    fake_eps = gen_fn(z_eps, fake_labels, GP, GB, bn_mode="track") if z_eps is not None else None
    x = AR.diffaug(fake, hp["diffaug_policy"], draws["series_fake"]) if hp.get("diffaug_policy") else fake
    adv, h = dis_fn(x, fake_labels, DP, DB)
    loss = g_loss(loss_kind, adv)
    if hp.get("fm_lambda") is not None:      # worker.py:588-596: a real batch through the (series-augmented) discriminator, its pooled features detached
        real, real_labels = hp["_fm_real"]
        xr = AR.diffaug(real, hp["diffaug_policy"], draws["series_real_fm"]) if hp.get("diffaug_policy") else real
        _, h_r = dis_fn(xr, real_labels, DP, DB)
        loss = loss + hp["fm_lambda"] * feature_matching(h_r.detach(), h)
    if hp.get("g_lambda") is not None:
        loss = loss - hp["g_lambda"] * ((fake - fake_eps) ** 2).mean()
    return loss, fake
