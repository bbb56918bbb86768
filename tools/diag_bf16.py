"""Diagnostic: single ops in bf16 mode vs torch with bf16 rounding at the assumed storage points (which op deviates from the
emulation model of oracle/restate.py Bf16Emu?). Prints relative-L2 errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as TF
import studiogan_amd
from studiogan_amd import ops, functional as F
from oracle import restate as O

dev = torch.device("cuda:0")
E = O.Bf16Emu
r = O._r


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def run_conv(cin, cout, k, stride, pad, hw, n=4, sn=True, seed=1, in_relu=False):
    torch.manual_seed(seed)
    ops.COMPUTE_DTYPE = torch.bfloat16
    m = (ops.snconv2d if sn else ops.conv2d)(cin, cout, k, stride, pad).to(dev)
    m.train()
    with torch.no_grad():
        m.bias.copy_(0.1 * torch.randn(cout))
    x = r(rnd(n, cin, hw, hw, seed=seed + 1))
    gy_shape = None
    P = {("c." + kk): v.detach().cpu().clone() for kk, v in m.named_parameters()}
    B = {("c." + kk): v.detach().cpu().clone() for kk, v in m.named_buffers()}
    xd = x.to(dev).requires_grad_(True)
    y = m(xd)
    gy = r(rnd(*y.shape, seed=seed + 2))
    y.backward(gy.to(dev).to(y.dtype))
    # emulation
    leaves = {kk: v.clone().requires_grad_(True) for kk, v in P.items()}
    xo = x.clone().requires_grad_(True)
    xin = torch.relu(xo) if in_relu else xo
    yo = E.q(O.conv_strided(xin, leaves, B, "c", stride, pad, True, E))
    yo.backward(gy)
    wname = "c.weight_orig" if sn else "c.weight"
    print(f"conv {cin}->{cout} k{k} s{stride} @{hw} sn={sn}: y {rel(y.float(), yo):.2e}  dx {rel(xd.grad, xo.grad):.2e}  dW {rel(dict(m.named_parameters())[wname[2:]].grad, leaves[wname].grad):.2e}  db {rel(m.bias.grad, leaves['c.bias'].grad):.2e}")


def run_bn(c, hw, n=8, relu=True, seed=3):
    torch.manual_seed(seed)
    ops.COMPUTE_DTYPE = torch.bfloat16
    m = ops.batchnorm_2d(c).to(dev)
    m.train()
    with torch.no_grad():
        m.weight.copy_(1 + 0.2 * torch.randn(c)); m.bias.copy_(0.2 * torch.randn(c))
    x = r(rnd(n, c, hw, hw, seed=seed))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).requires_grad_(True)
    y = m.forward_nhwc(xd, relu=relu)
    gy = r(rnd(n, hw, hw, c, seed=seed + 1))
    y.backward(gy.to(dev).to(torch.bfloat16))
    w, b = m.weight.detach().cpu().clone().requires_grad_(True), m.bias.detach().cpu().clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    t = TF.batch_norm(E.qb(xo), None, None, w, b, True, 0.1, 1e-4)
    yo = E.q(torch.relu(t) if relu else t)
    yo.backward(gy.permute(0, 3, 1, 2))
    print(f"bn c{c} @{hw} relu={relu}: y {rel(y.float().permute(0, 3, 1, 2), yo):.2e}  dx {rel(xd.grad.float().permute(0, 3, 1, 2), xo.grad):.2e}  dgain {rel(m.weight.grad, w.grad):.2e}  dbias {rel(m.bias.grad, b.grad):.2e}")


class _MOD:
    pass


def run_attn(c, hw, n=2, seed=5):
    torch.manual_seed(seed)
    ops.COMPUTE_DTYPE = torch.bfloat16
    MOD = ops.Modules(apply_g_sn=True, apply_d_sn=True, g_cond_mtd="cBN", backbone="big_resnet")
    m = ops.SelfAttention(c, True, MOD).to(dev)
    m.compute_dtype = torch.bfloat16
    ops.adopt(m, torch.bfloat16)
    m.train()
    with torch.no_grad():
        m.sigma.fill_(0.6)
    P = {("a." + kk): v.detach().cpu().clone() for kk, v in m.named_parameters()}
    B = {("a." + kk): v.detach().cpu().clone() for kk, v in m.named_buffers()}
    x = r(rnd(n, c, hw, hw, seed=seed))
    from studiogan_amd.bank import get_bank
    bank = get_bank(m, torch.bfloat16)
    slot = bank.begin_forward(True)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).requires_grad_(True)
    y = m.forward_nhwc(xd, slot)
    gy = r(rnd(n, hw, hw, c, seed=seed + 1))
    y.backward(gy.to(dev).to(torch.bfloat16))
    leaves = {kk: v.clone().requires_grad_(True) for kk, v in P.items()}
    xo = x.clone().requires_grad_(True)
    yo = O.self_attention(xo, leaves, B, "a", True, E)
    yo.backward(gy.permute(0, 3, 1, 2))
    print(f"attn c{c} @{hw}: y {rel(y.float().permute(0, 3, 1, 2), yo):.2e}  dx {rel(xd.grad.float().permute(0, 3, 1, 2), xo.grad):.2e}", end="")
    for kk, p in m.named_parameters():
        print(f"  d{kk.split('.')[0][-5:]} {rel(p.grad, leaves['a.' + kk].grad):.2e}", end="")
    print()


if __name__ == "__main__":
    run_conv(16, 32, 3, 1, 1, 16)
    run_conv(64, 64, 3, 1, 1, 16, sn=False)
    run_conv(128, 256, 3, 1, 1, 8)
    run_conv(64, 64, 4, 2, 1, 16, sn=False)
    run_conv(3, 64, 3, 1, 1, 32, sn=False)
    run_bn(64, 16)
    run_bn(512, 4, relu=False)
    run_attn(64, 16)
    run_attn(192, 32)
