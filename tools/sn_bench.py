"""Spectral-norm launches alone (csrc/sn.hip) on the weights of the benchmarked networks (BigGAN-128 G and D, ch 96, bf16 images): one forward
(power iteration + sigma + operand images) and one backward (dW = (dWt - <dWt, W/sigma> u v^T) / sigma into the gradient arena) per network.
    python tools/sn_bench.py [--iters 20]
GB/s = algorithmic bytes (the accounting of sg_sn_forward / sg_sn_backward: every pass reads / writes its tensors once) / hipEvent time."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from studiogan_amd.bank import get_bank  # noqa: E402


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
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    G, D = bench.build(bench.WORKLOADS["biggan128"], True, dev)
    print(f"SG_SN_CHUNK_MB={os.environ.get('SG_SN_CHUNK_MB', '(default)')}")
    for name, net in (("G", G), ("D", D)):
        net.train()
        bank = get_bank(net, torch.bfloat16)
        elems = sum(r.rows * r.cols for r in bank.layers)
        sn_elems = sum(r.rows * r.cols for r in bank.layers if r.apply_sn)
        conv_elems = sum(r.rows * r.cols for r in bank.layers if r.kind == "conv")
        f32_elems = sum(r.rows * r.cols for r in bank.layers if r.want_f32)
        holder = []

        def fwd():
            holder[:] = [bank.begin_forward(True)]
        t_f = timeit(fwd, args.iters)
        by_f = sn_elems * 8.0 + elems * 4.0 + conv_elems * 4.0 + f32_elems * 4.0      # W^T u + W v, pack read, two bf16 images / fp32 copy
        slot = bank.begin_forward(True)

        def bwd():
            bank._cb_queued = True          # (no autograd pass here: the flush below is called by hand)
            for r in bank.layers:
                bank.dwt(slot, r)
            bank.flush()
        t_b = timeit(bwd, args.iters)
        by_b = sn_elems * 8.0 + elems * 12.0
        print(f"{name}: {len(bank.layers)} layers, {elems / 1e6:.1f} M weights | forward {t_f * 1e3:7.1f} us = {by_f / t_f / 1e6:7.1f} GB/s ({by_f / t_f / 8e6 * 100 / 1e3:.1f} % of 8 TB/s) | "
              f"backward {t_b * 1e3:7.1f} us = {by_b / t_b / 1e6:7.1f} GB/s ({by_b / t_b / 8e6 * 100 / 1e3:.1f} % of 8 TB/s)")


if __name__ == "__main__":
    main()
