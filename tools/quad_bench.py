"""Quad convolutions (csrc/conv_q.h, wgrad_q.h) against the direct 3x3 kernels on the BigGAN-128 (C3) layers that sit next to a 2x resampling,
batch 256, bf16: forward, data gradient, weight gradient. TF = algorithmic (the 3x3 convolution over the fine grid) / time; the quad
launches execute 16/36 of those FLOPs.
    python tools/quad_bench.py [--batch 256] [--only 192-192-32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import studiogan_amd  # noqa: E402,F401
from studiogan_amd import functional as F, _lib as L  # noqa: E402

# (form, Cin, Cout, Hl): POOL = D block tails (conv at 2 Hl, pooled to Hl), UP = G block heads (source Hl, conv at 2 Hl)
SHAPES = [
    ("pool", 96, 96, 64), ("pool", 192, 192, 32), ("pool", 384, 384, 16), ("pool", 768, 768, 8), ("pool", 1536, 1536, 4),
    ("up", 1536, 1536, 4), ("up", 1536, 768, 8), ("up", 768, 384, 16), ("up", 384, 192, 32), ("up", 192, 96, 64),
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
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev, dt, N = torch.device("cuda:0"), torch.bfloat16, args.batch
    print(f"{'layer':26s} {'GFLOP':>7s} | {'fwd 3x3':>8s} {'quad':>7s} {'TF':>6s} {'x':>5s} | {'dgrad 3x3':>9s} {'quad':>7s} {'TF':>6s} {'x':>5s} | {'wgrad 3x3':>9s} {'quad':>7s} {'TF':>6s} {'x':>5s}")
    tot = [0.0] * 7
    for form, Cin, Cout, Hl in SHAPES:
        if args.only and f"{Cin}-{Cout}-{Hl}" not in args.only.split(","):
            continue
        pool = form == "pool"
        Hf = 2 * Hl
        Hx = Hf if pool else Hl
        Hy = Hl if pool else Hf
        x = torch.randn(N, Hx, Hx, Cin, device=dev).to(dt)
        gy = torch.randn(N, Hy, Hy, Cout, device=dev).to(dt)
        w = (0.05 * torch.randn(Cout, 3, 3, Cin, device=dev)).to(dt)
        wd = w.flip(1).flip(2).permute(3, 1, 2, 0).contiguous()
        bias = torch.randn(Cout, device=dev)
        qf = torch.empty(Cout, 16, Cin, dtype=dt, device=dev)
        qd = torch.empty(Cin, 16, Cout, dtype=dt, device=dev)
        F.quad_pack_raw(w.data_ptr(), qf, 0 if pool else 1, Cout, Cin)
        F.quad_pack_raw(wd.data_ptr(), qd, 2 if pool else 3, Cin, Cout)
        dw = torch.zeros(Cout, 9, Cin, device=dev)
        relu = pool      # D tails read relu(h); G heads read the BN output (ReLU fused upstream)
        use_mask = relu and os.environ.get("SG_QB_NOMASK") != "1"      # SG_QB_NOMASK=1: the data gradient without its ReLU-mask operand (what the mask tile costs)
        pf = (L.PIX_RELU if relu else 0) | (0 if pool else L.PIX_UPSAMPLE)
        ef = L.EPI_POOL if pool else 0
        qform, dform = (L.Q_POOL, L.Q_UP) if pool else (L.Q_UP, L.Q_POOL)
        flop = 2.0 * N * Hf * Hf * Cout * 9 * Cin
        f0 = timeit(lambda: F.conv2d_raw(x, w.data_ptr(), Cin, Cout, 3, 3, 1, 1, 1, pf, ef, bias=bias, alpha=0.25 if pool else 1.0))
        f1 = timeit(lambda: F.conv2d_q_raw(x, qf.data_ptr(), qform, Cin, Cout, L.PIX_RELU if relu else 0, 0, bias=bias))
        d0 = timeit(lambda: F.conv2d_raw(gy, wd.data_ptr(), Cout, Cin, 3, 3, 1, 1, 1, L.PIX_UPSAMPLE if pool else 0, 0 if pool else L.EPI_POOL,
                                         mask=x if use_mask else None, alpha=0.25 if pool else 1.0))
        d1 = timeit(lambda: F.conv2d_q_raw(gy, qd.data_ptr(), dform, Cout, Cin, 0, 0, mask=x if use_mask else None))
        g0 = timeit(lambda: F.conv2d_wgrad_raw(x, gy, dw.data_ptr(), Cin, Cout, 3, 3, Hf, Hf, 1, 1, 1, pf, L.PIX_UPSAMPLE if pool else 0,
                                               alpha=0.25 if pool else 1.0))
        g1 = timeit(lambda: F.conv2d_q_wgrad_raw(x, gy, dw.data_ptr(), qform, Cin, Cout, L.PIX_RELU if relu else 0))
        tf = lambda ms: flop / ms / 1e9
        print(f"{form:4s} {Cin:5d}->{Cout:5d} @{Hl:3d}^2 {flop / 1e9:7.1f} | {f0:8.3f} {f1:7.3f} {tf(f1):6.0f} {f0 / f1:5.2f} | {d0:9.3f} {d1:7.3f} {tf(d1):6.0f} {d0 / d1:5.2f} | "
              f"{g0:9.3f} {g1:7.3f} {tf(g1):6.0f} {g0 / g1:5.2f}")
        for k, v in enumerate((flop, f0, f1, d0, d1, g0, g1)):
            tot[k] += v
    if tot[0]:
        tf = lambda ms: tot[0] / ms / 1e9
        print(f"{'sum':26s} {tot[0] / 1e9:7.1f} | {tot[1]:8.3f} {tot[2]:7.3f} {tf(tot[2]):6.0f} {tot[1] / tot[2]:5.2f} | {tot[3]:9.3f} {tot[4]:7.3f} {tf(tot[4]):6.0f} {tot[3] / tot[4]:5.2f} | "
              f"{tot[5]:9.3f} {tot[6]:7.3f} {tf(tot[6]):6.0f} {tot[5] / tot[6]:5.2f}")


if __name__ == "__main__":
    main()
