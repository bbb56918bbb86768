"""Fused 3x3 + 1x1-skip launch (csrc/conv_v4.h SKIP) against the two launches it replaces, on the residual-block tails of BigGAN-128 (C3) at
batch 256, bf16.   python tools/skip_bench.py [--batch 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import studiogan_amd  # noqa: E402,F401
from studiogan_amd import functional as F, _lib as L  # noqa: E402
from conv_bench import timeit  # noqa: E402

# (C main, Cout, C2 skip, H out, relu, pool, up2): D blocks 1-4 (pooled), G blocks 1-5 (skip input at half resolution)
SHAPES = [(192, 192, 96, 64, True, True, False), (384, 384, 192, 32, True, True, False), (768, 768, 384, 16, True, True, False),
          (1536, 1536, 768, 8, True, True, False),
          (1536, 1536, 1536, 8, False, False, True), (768, 768, 1536, 16, False, False, True), (384, 384, 768, 32, False, False, True),
          (192, 192, 384, 64, False, False, True), (96, 96, 192, 128, False, False, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    dev, dt, N = torch.device("cuda:0"), torch.bfloat16, args.batch
    print(f"{'block tail':44s} {'GFLOP':>8s} | {'3x3 ms':>7s} {'1x1 ms':>7s} {'sum':>7s} | {'fused ms':>8s} {'TF':>7s} | saved")
    tot = [0.0, 0.0]
    for (C, Cout, C2, H, relu, pool, up2) in SHAPES:
        H2 = H // 2 if up2 else H
        h = torch.randn(N, H, H, C, device=dev).to(dt)
        x = torch.randn(N, H2, H2, C2, device=dev).to(dt)
        w = (0.05 * torch.randn(Cout, 3, 3, C, device=dev)).to(dt)
        w0 = (0.05 * torch.randn(Cout, C2, device=dev)).to(dt)
        b, b0 = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
        pf = L.PIX_RELU if relu else 0
        ef = L.EPI_POOL if pool else 0
        al = 0.25 if pool else 1.0
        flop = 2.0 * N * H * H * Cout * (9 * C + C2)
        hh = F.conv2d_raw(h, w.data_ptr(), C, Cout, 3, 3, 1, 1, 1, pf, ef, bias=b, alpha=al)
        t3 = timeit(lambda: F.conv2d_raw(h, w.data_ptr(), C, Cout, 3, 3, 1, 1, 1, pf, ef, bias=b, alpha=al))
        t1 = timeit(lambda: F.conv2d_raw(x, w0.data_ptr(), C2, Cout, 1, 1, 1, 0, 0, pf | (L.PIX_UPSAMPLE if up2 else 0), ef, bias=b0, res=hh, alpha=al))
        y = F.conv2d_skip_raw(h, w.data_ptr(), C, Cout, x, w0.data_ptr(), C2, up2, pf, ef, bias=b, bias2=b0, alpha=al)
        tf = timeit(lambda: F.conv2d_skip_raw(h, w.data_ptr(), C, Cout, x, w0.data_ptr(), C2, up2, pf, ef, bias=b, bias2=b0, alpha=al)) if y is not None else float("nan")
        name = f"{C}->{Cout} @{H}^2 + skip {C2}{' up' if up2 else ''}{' relu,pool' if pool else ''}"
        print(f"{name:44s} {flop / 1e9:8.1f} | {t3:7.3f} {t1:7.3f} {t3 + t1:7.3f} | {tf:8.3f} {flop / tf / 1e9:7.1f} | {t3 + t1 - tf:+.3f} ms")
        tot[0] += t3 + t1
        tot[1] += tf
    print(f"{'sum':44s} {'':8s} | {'':7s} {'':7s} {tot[0]:7.3f} | {tot[1]:8.3f}")


if __name__ == "__main__":
    main()
