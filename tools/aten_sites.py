"""Which torch (aten) operators launch device work during one C3 training step, and from where: every such launch is a kernel boundary on a stream that is
busy end to end (~8 us each, measured by removing ~110 of them: profiles/README_r06.md). torch.profiler with Python stacks over ONE step of bench.py's worker;
rows = (operator, first frame inside the package), sorted by launches.
    python tools/aten_sites.py [--workload biggan128] [--batch 256]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="biggan128")
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    from studiogan_amd.worker import Worker
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[args.workload]
    torch.manual_seed(1234)
    G, D = bench.build(wl, True, dev)
    w = Worker(G, D, wl["z_dim"], wl["classes"], args.batch, "hinge", wl["g_lr"], wl["d_lr"], wl["beta1"], wl["beta2"],
               d_updates_per_step=wl["n_d"], apply_g_ema=True, g_ema_decay=0.9999, g_ema_start=20000)
    n_d = wl["n_d"]
    pool = bench.generator_real_pool(G, n_d * 5, args.batch, wl["z_dim"], wl["classes"], dev, 1234)
    for i in range(3):
        w.step(i, bench.baskets(pool, i, n_d))
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        w.step(3, bench.baskets(pool, 3, n_d))
        torch.cuda.synchronize()
    rows = collections.Counter()
    for e in prof.events():
        if not e.kernels:
            continue
        if any(c.kernels for c in (e.cpu_children or [])):      # count the innermost operator that owns the launch
            continue
        site = "?"
        for fr in (e.stack or []):
            if "studiogan_amd" in fr and "torch/" not in fr:
                site = fr.split("studiogan_amd/")[-1]
                break
        else:
            for fr in (e.stack or []):
                if "bench.py" in fr:
                    site = fr.split("/")[-1]
                    break
        rows[(e.name, site)] += len(e.kernels)
    tot = sum(rows.values())
    print(f"{tot} device launches from torch operators in one step")
    for (name, site), n in rows.most_common(70):
        print(f"{n:5d}  {name:40s} {site}")


if __name__ == "__main__":
    main()
