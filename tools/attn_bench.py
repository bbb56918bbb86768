"""Self-attention core alone (csrc/attn.hip through functional.AttnCoreFn: max-pool of phi / g, scores, softmax, P.V and the fused backward) at the
benchmarked shapes, bf16:
    python tools/attn_bench.py [--iters 10]
G of C3: batch 256, 64 x 64, 24 (padded 32) -> 96 channels; D of C3: 64 x 64, 12 (16) -> 48; D of C4 at 256^2: batch 64, 128 x 128, 32 -> 128.
TF = MFMA FLOPs of the formulation (forward: scores twice -- max pass + main pass -- and P.V; backward: scores and dP on both sides, dtheta, dphi, dg) / time."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import studiogan_amd  # noqa: E402,F401
from studiogan_amd import functional as F  # noqa: E402

SHAPES = [("C3 G", 256, 64, 32, 96), ("C3 D", 256, 64, 16, 48), ("C4-256 D", 64, 128, 32, 128)]


def timeit(fn, iters):
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
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    print(f"{'shape':10s} {'B':>4s} {'HW':>6s} {'Dp':>3s} {'Cg':>4s} | {'fwd (no grad) ms':>17s} {'TF':>7s} | {'fwd + bwd ms':>13s} {'TF':>7s}")
    for name, B, H, Dp, Cg in SHAPES:
        HW, HW4 = H * H, H * H // 4
        th = (0.5 * torch.randn(B, H, H, Dp, device=dev)).to(dt)
        ph = (0.5 * torch.randn(B, H, H, Dp, device=dev)).to(dt)
        g = torch.randn(B, H, H, Cg, device=dev).to(dt)
        go = torch.randn(B, H, H, Cg, device=dev).to(dt)

        def fwd():
            with torch.no_grad():
                F.AttnCoreFn.apply(th, ph, g)

        thg, phg, gg = th.clone().requires_grad_(True), ph.clone().requires_grad_(True), g.clone().requires_grad_(True)

        def fwd_bwd():
            o = F.AttnCoreFn.apply(thg, phg, gg)
            o.backward(go)
            thg.grad = phg.grad = gg.grad = None
        tf_ = timeit(fwd, args.iters)
        tb_ = timeit(fwd_bwd, args.iters)
        blk = 2.0 * B * HW * HW4                   # one [HW x HW4] x 1-channel product
        f_fwd = blk * (2 * Dp + Cg)
        f_bwd = blk * (2 * (Dp + Cg) + Dp + Dp + Cg)
        print(f"{name:10s} {B:4d} {HW:6d} {Dp:3d} {Cg:4d} | {tf_:17.3f} {f_fwd / tf_ / 1e9:7.1f} | {tb_:13.3f} {(f_fwd + f_bwd) / tb_ / 1e9:7.1f}")


if __name__ == "__main__":
    main()
