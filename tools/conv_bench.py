"""Per-layer micro-benchmark of the convolution engine on the BigGAN-128 (C3) layer shapes at batch 256, bf16.
Prints achieved algorithmic TFLOP/s for forward, data gradient and weight gradient of each shape.
    python tools/conv_bench.py [--batch 256] [--fp32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import studiogan_amd  # noqa: E402,F401
from studiogan_amd import functional as F, _lib as L  # noqa: E402

# (Cin, Cout, H(out), R, flags) : the D stack of BigGAN-128 + the G-only variants
SHAPES = [
    (96, 96, 128, 3, ""), (96, 192, 64, 3, ""), (192, 192, 64, 3, ""), (192, 384, 32, 3, ""), (384, 384, 32, 3, ""),
    (384, 768, 16, 3, ""), (768, 768, 16, 3, ""), (768, 1536, 8, 3, ""), (1536, 1536, 8, 3, ""), (1536, 1536, 4, 3, ""),
    (96, 192, 64, 1, ""), (1536, 1536, 8, 1, "up"), (1536, 768, 16, 3, "up"), (192, 96, 128, 3, "up"), (96, 96, 128, 3, "relu,pool"),
    # the RGB layers as the networks run them: 8 padded channels on the thin side (zero-filled), FLOPs counted for the 3 real ones
    (8, 96, 128, 3, "rgb_in"), (96, 8, 128, 3, "rgb_out"), (8, 96, 64, 1, "rgb_in"),
    (96, 16, 64, 1, ""), (96, 48, 64, 1, ""), (48, 96, 64, 1, ""), (192, 384, 32, 1, ""), (192, 96, 128, 1, "up"),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--bias", action="store_true", help="forward with a bias vector (as every layer of the networks has)")
    args = ap.parse_args()
    dt = torch.float32 if args.fp32 else torch.bfloat16
    dev = torch.device("cuda:0")
    N = args.batch
    print(f"{'shape':34s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF':>7s} | {'dgrad ms':>8s} {'TF':>7s} | {'wgrad ms':>8s} {'TF':>7s}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for (Cin, Cout, H, R, fl) in SHAPES:
        if args.only and not any(o == f"{Cin}-{Cout}-{H}" or (o in f"{Cin}-{Cout}-{H}-{fl}" and "-" not in o.replace(f"{Cin}-{Cout}-{H}", "")) for o in args.only.split(",")):
            continue
        up, relu, pool = "up" in fl, "relu" in fl, "pool" in fl
        Hs = H // 2 if up else H
        pad = R // 2
        x = torch.randn(N, Hs, Hs, Cin, device=dev).to(dt)
        w = (0.05 * torch.randn(Cout, R, R, Cin, device=dev)).to(dt)
        wd = (0.05 * torch.randn(Cin, R, R, Cout, device=dev)).to(dt)
        Hy = H // 2 if pool else H
        gy = torch.randn(N, Hy, Hy, Cout, device=dev).to(dt)
        dw = torch.zeros(Cout, R, R, Cin, device=dev)
        pf = (L.PIX_UPSAMPLE if up else 0) | (L.PIX_RELU if relu else 0)
        ef = L.EPI_POOL if pool else 0
        flop = 2.0 * N * H * H * (3 if "rgb_out" in fl else Cout) * R * R * (3 if "rgb_in" in fl else Cin)
        bias = torch.randn(Cout, device=dev) if args.bias else None
        f = timeit(lambda: F.conv2d_raw(x, w.data_ptr(), Cin, Cout, R, R, 1, pad, pad, pf, ef, bias=bias, alpha=0.25 if pool else 1.0))
        d = timeit(lambda: F.conv2d_raw(gy, wd.data_ptr(), Cout, Cin, R, R, 1, pad, pad, L.PIX_UPSAMPLE if pool else 0, L.EPI_POOL if up else 0,
                                        mask=x if relu else None, alpha=0.25 if pool else 1.0))
        g = timeit(lambda: F.conv2d_wgrad_raw(x, gy, dw.data_ptr(), Cin, Cout, R, R, H, H, 1, pad, pad, pf, L.PIX_UPSAMPLE if pool else 0))
        print(f"{Cin:5d}->{Cout:5d} @{H:3d}^2 k{R} {fl:10s} {flop / 1e9:8.1f} | {f:8.3f} {flop / f / 1e9:7.1f} | {d:8.3f} {flop / d / 1e9:7.1f} | {g:8.3f} {flop / g / 1e9:7.1f}")
        tot[0] += flop; tot[1] += f; tot[2] += d; tot[3] += g
    print(f"{'sum':34s} {tot[0] / 1e9:8.1f} | {tot[1]:8.3f} {tot[0] / tot[1] / 1e9:7.1f} | {tot[2]:8.3f} {tot[0] / tot[2] / 1e9:7.1f} | {tot[3]:8.3f} {tot[0] / tot[3] / 1e9:7.1f}")


if __name__ == "__main__":
    main()
