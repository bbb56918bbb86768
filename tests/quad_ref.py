"""Index-level CPU restatement of the quad convolutions (csrc/conv_q.h, wgrad_q.h, conv_q.hip): the same view / tap / origin conventions
as the kernels, written with plain tensor slicing. tests/test_quad_cpu.py checks it against torch's conv2d + avg_pool2d / interpolate
(exact in fp64) -- so the conventions the HIP kernels implement are pinned on the CPU -- and tests/test_quad_gpu.py compares the kernels
with it and with torch. Test infrastructure only."""
import torch
import torch.nn.functional as F_

PAT_P = [[0b011, 0b100], [0b001, 0b110]]   # POOL-like: 3x3 tap indices r feeding quad tap t of parity a (conv_q.hip quad_pat)
PAT_U = [[0b001, 0b110], [0b011, 0b100]]   # UP-like


def _pat(mode):
    return (PAT_P if mode in (0, 3) else PAT_U), (0.25 if mode in (0, 2) else 1.0)


def quad_pack_ref(w9, mode):
    """w9: [M, 3, 3, Cs] -> [M, 4 views, 4 taps, Cs] (sg_quad_pack)"""
    pat, scale = _pat(mode)
    M, _, _, Cs = w9.shape
    out = torch.zeros(M, 4, 4, Cs, dtype=w9.dtype)
    for view in range(4):
        for t in range(4):
            pr, pc = pat[view >> 1][t >> 1], pat[view & 1][t & 1]
            for r in range(3):
                for s in range(3):
                    if (pr >> r) & 1 and (pc >> s) & 1:
                        out[:, view, t] += w9[:, r, s]
    return out * scale


def quad_fold_ref(dq, form):
    """dq: [M, 4, 4, Cs] gradient w.r.t. the forward quad image of `form` (0 POOL, 1 UP) -> [M, 3, 3, Cs] (k_quad_reduce_fold)"""
    pat, scale = _pat(form)
    M, _, _, Cs = dq.shape
    out = torch.zeros(M, 3, 3, Cs, dtype=dq.dtype)
    for view in range(4):
        for t in range(4):
            pr, pc = pat[view >> 1][t >> 1], pat[view & 1][t & 1]
            for r in range(3):
                for s in range(3):
                    if (pr >> r) & 1 and (pc >> s) & 1:
                        out[:, r, s] += dq[:, view, t]
    return out * scale


def flipped_transposed(w9):
    """[Cout, 3, 3, Cin] -> the 3x3 data-gradient image [Cin, 2-r, 2-s, Cout] (csrc/sn.hip k_sn_pack_dgrad)"""
    return w9.flip(1).flip(2).permute(3, 1, 2, 0).contiguous()


def _shift(x, di, dj):
    """y[n, i, j] = x[n, i + di, j + dj] (zeros outside), NHWC"""
    N, H, W, C = x.shape
    y = torch.zeros_like(x)
    i0, i1 = max(0, -di), min(H, H - di)
    j0, j1 = max(0, -dj), min(W, W - dj)
    if i1 > i0 and j1 > j0:
        y[:, i0:i1, j0:j1] = x[:, i0 + di:i1 + di, j0 + dj:j1 + dj]
    return y


def convq_ref(x, wq, form):
    """x NHWC; wq [Cout, 4, 4, C]. form 0 (POOL): x fine -> out low; form 1 (UP): x low -> out fine. (sg_conv_q_kernel)"""
    Cout = wq.shape[0]
    if form == 0:
        N, H2, W2, C = x.shape
        out = torch.zeros(N, H2 // 2, W2 // 2, Cout, dtype=x.dtype)
        for view in range(4):
            a, b = view >> 1, view & 1
            xv = x[:, a::2, b::2]
            for t in range(4):
                ti, tj = t >> 1, t & 1
                out += torch.einsum("nijc,oc->nijo", _shift(xv, ti - a, tj - b), wq[:, view, t])
        return out
    N, H, W, C = x.shape
    out = torch.zeros(N, 2 * H, 2 * W, Cout, dtype=x.dtype)
    for view in range(4):
        a, b = view >> 1, view & 1
        acc = torch.zeros(N, H, W, Cout, dtype=x.dtype)
        for t in range(4):
            ti, tj = t >> 1, t & 1
            acc += torch.einsum("nijc,oc->nijo", _shift(x, ti - (1 - a), tj - (1 - b)), wq[:, view, t])
        out[:, a::2, b::2] = acc
    return out


def wgradq_ref(x, dy, form):
    """gradient w.r.t. the forward quad image: [Cout, 4, 4, C] (sg_wgrad_q_kernel)"""
    C, Cout = x.shape[3], dy.shape[3]
    dq = torch.zeros(Cout, 4, 4, C, dtype=x.dtype)
    for view in range(4):
        a, b = view >> 1, view & 1
        for t in range(4):
            ti, tj = t >> 1, t & 1
            if form == 0:
                xs, g = _shift(x[:, a::2, b::2], ti - a, tj - b), dy
            else:
                xs, g = _shift(x, ti - (1 - a), tj - (1 - b)), dy[:, a::2, b::2]
            dq[:, view, t] = torch.einsum("nijo,nijc->oc", g, xs)
    return dq


def pool_conv_torch(x, w9, relu=False):
    """avgpool2(conv3x3(relu?(x))) with torch ops; x NHWC, w9 [Cout,3,3,C] -> NHWC"""
    xx = x.permute(0, 3, 1, 2)
    if relu:
        xx = torch.relu(xx)
    y = F_.avg_pool2d(F_.conv2d(xx, w9.permute(0, 3, 1, 2), padding=1), 2)
    return y.permute(0, 2, 3, 1)


def up_conv_torch(x, w9, relu=False):
    xx = x.permute(0, 3, 1, 2)
    if relu:
        xx = torch.relu(xx)
    y = F_.conv2d(F_.interpolate(xx, scale_factor=2, mode="nearest"), w9.permute(0, 3, 1, 2), padding=1)
    return y.permute(0, 2, 3, 1)
