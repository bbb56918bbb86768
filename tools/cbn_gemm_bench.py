"""The tiny fp32 GEMMs of the conditional batch norms and of linear0 (functional.CbnAffineFn / LinearFn) at BigGAN-128's shapes, batch 256: forward
[B x K] x [K x 2C], data gradient (one launch / contraction sliced over the batches of one launch + fixed-order sum: functional.gemm_dgrad_rows), weight gradient (contraction over the batch). These are latency-bound launches (3.5 ms of the C3 step in ~130 launches).
    python tools/cbn_gemm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import studiogan_amd  # noqa: E402,F401
from studiogan_amd import functional as F, _lib as L  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    B, K = 256, 148
    print(f"{'layer':28s} {'fwd us':>8s} {'dgrad us':>9s} {'sliced us':>9s} | weight gradient us at splits = 1, 2, 4, 8 (> 1: atomic epilogue, measurement only)")
    for name, rows, K in [("cBN 2C=3072", 3072, 148), ("cBN 2C=1536", 1536, 148), ("cBN 2C=768", 768, 148), ("cBN 2C=384", 384, 148), ("cBN 2C=192", 192, 148),
                          ("linear0 24576 x 20", 24576, 20)]:
        W = torch.randn(rows, K, device=dev)
        y = torch.randn(B, K, device=dev)
        out = torch.empty(B, rows, device=dev)
        dgb = torch.randn(B, rows, device=dev)
        dy = torch.empty(B, K, device=dev)
        dW = torch.zeros(rows, K, device=dev)
        t_f = timeit(lambda: F.gemm_raw(L.F32, W, 0, K, y, 0, K, out, rows, rows, B, K))
        t_d = timeit(lambda: F.gemm_raw(L.F32, W, 1, K, dgb, 0, rows, dy, K, K, B, rows))
        t_ds = timeit(lambda: F.gemm_dgrad_rows(W.data_ptr(), dgb, dy, B, K, rows))
        tw = []
        for sp in (1, 2, 4, 8):
            try:
                tw.append(timeit(lambda: F.gemm_raw(L.F32, y, 1, K, dgb, 1, rows, dW, K, K, rows, B, splits=sp, epi_flags=L.EPI_ATOMIC if sp > 1 else 0)))
            except RuntimeError as e:
                tw.append(float("nan"))
        print(f"{name:28s} {t_f:8.1f} {t_d:9.1f} {t_ds:9.1f} | " + "  ".join(f"{t:8.1f}" for t in tw))


if __name__ == "__main__":
    main()
