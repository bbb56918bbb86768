"""Batch-norm apply pass alone (csrc/norm.hip sg_bn_apply: y = relu(((x - mean) * invstd) * gain[n] + bias[n]), bf16, the conditional form of BigGAN's generator) on the
activation shapes of BigGAN-128's generator at batch 256, per SG_BN_APPLY variant ("<variant><blocks / 1024>", read once per process: this script re-runs itself per value).
    python tools/bn_bench.py [--variants 02,12,22,32,04,14]
GB/s = (read + write of the activation) / hipEvent time."""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(256, 128 * 128, 96), (256, 64 * 64, 192), (256, 32 * 32, 384), (256, 16 * 16, 768), (256, 8 * 8, 1536), (256, 4 * 4, 1536)]


def child():
    import torch
    import studiogan_amd  # noqa: F401
    from studiogan_amd import _lib as L
    dev = torch.device("cuda:0")
    tot = 0.0
    line = []
    for (N, HW, C) in SHAPES:
        x = torch.randn(N, HW, C, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x)
        mean, invstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        gain, bias = torch.randn(N, C, device=dev), torch.randn(N, C, device=dev)

        def fn():
            L.call("sg_bn_apply", L.dt(torch.bfloat16), x.data_ptr(), y.data_ptr(), N, HW, C, mean.data_ptr(), invstd.data_ptr(), gain.data_ptr(), bias.data_ptr(), C, 1, L.stream())
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        tot += us
        line.append(f"{us:7.1f} us {2.0 * x.numel() * 2 / us / 1e3:6.0f} GB/s")
    print(f"SG_BN_APPLY={os.environ.get('SG_BN_APPLY', '(default)'):10s} | " + " | ".join(line) + f" | sum {tot:7.1f} us")


if __name__ == "__main__":
    if os.environ.get("SG_BN_BENCH_CHILD") == "1":
        child()
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("--variants", default="02,12,22,32,04,14,24,34")
        args = ap.parse_args()
        print("shapes (N, HW, C): " + " ".join(str(s) for s in SHAPES))
        for v in args.variants.split(","):
            env = dict(os.environ, SG_BN_BENCH_CHILD="1", SG_BN_APPLY=v)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith("SG_BN_APPLY")]
            print(out[0] if out else ("failed: " + r.stderr[-300:]))
