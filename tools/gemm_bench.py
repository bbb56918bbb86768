"""Micro-benchmark of the contraction engine's four operand-form combinations (bf16), to separate LDS/fragment-path
costs from the convolution gather costs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import studiogan_amd  # noqa: F401
from studiogan_amd import functional as F, _lib as L


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


dev = torch.device("cuda:0")
for (I, J, K) in [(2048, 4096, 4096), (1536, 4096, 13824)]:
    for dt in (torch.bfloat16, torch.float32):
        for pf, qf in [(0, 0), (0, 1), (1, 0), (1, 1)]:
            P = torch.randn((I, K) if pf == 0 else (K, I), device=dev).to(dt)
            Q = torch.randn((J, K) if qf == 0 else (K, J), device=dev).to(dt)
            out = torch.empty((J, I), device=dev, dtype=torch.float32)
            for no_tr in ([0, 1] if (dt == torch.bfloat16 and (pf or qf)) else [0]):
                ms = timeit(lambda: F.gemm_raw(L.dt(dt), P, pf, P.shape[1], Q, qf, Q.shape[1], out, I, I, J, K, epi_flags=L.EPI_OUT_F32, no_tr=no_tr))
                print(f"{str(dt):15s} I={I} J={J} K={K} p{pf}q{qf} no_tr={no_tr}: {ms:8.3f} ms {2.0 * I * J * K / ms / 1e9:8.1f} TF")
