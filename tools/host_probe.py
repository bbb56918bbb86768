"""Where does the HOST spend a training step? bench.py's host_lead shows the first timed steps after a synchronisation issued in ~110 ms each and later ones in ~25 ms (the host then runs
into the runtime's cap on outstanding work). This probe wraps studiogan_amd._lib.call and reports, per step: host time, time inside C-ABI calls, the calls that took longest.
    python tools/host_probe.py [--steps 8] [--warmup 3]"""
import argparse
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    from studiogan_amd import _lib as L
    from studiogan_amd.worker import Worker
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["biggan128"]
    torch.manual_seed(1234)
    G, D = bench.build(wl, True, dev)
    w = Worker(G, D, wl["z_dim"], wl["classes"], 256, "hinge", wl["g_lr"], wl["d_lr"], wl["beta1"], wl["beta2"], d_updates_per_step=wl["n_d"], apply_g_ema=True, g_ema_decay=0.9999, g_ema_start=20000)
    n_d = wl["n_d"]
    pool = bench.generator_real_pool(G, n_d * 6, 256, wl["z_dim"], wl["classes"], dev, 1234)
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    orig = L.call

    def timed(name, *a):
        t = time.perf_counter()
        try:
            return orig(name, *a)
        finally:
            dt = time.perf_counter() - t
            e = acc[name]
            e[0] += 1; e[1] += dt; e[2] = max(e[2], dt)
    L.call = timed
    for i in range(args.warmup):
        w.step(i, bench.baskets(pool, i, n_d))
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    evs = []
    for i in range(args.steps):
        acc.clear()
        ts = time.perf_counter()
        w.step(args.warmup + i, bench.baskets(pool, args.warmup + i, n_d))
        host = time.perf_counter() - ts
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((e, time.perf_counter() - t0))
        inside = sum(v[1] for v in acc.values())
        top = sorted(acc.items(), key=lambda kv: -kv[1][1])[:4]
        print(f"step {i}: host {host * 1e3:7.1f} ms, inside C-ABI calls {inside * 1e3:7.1f} ms ({sum(v[0] for v in acc.values())} calls) | " +
              " | ".join(f"{k} n={v[0]} {v[1] * 1e3:.1f} ms max {v[2] * 1e3:.2f}" for k, v in top))
    torch.cuda.synchronize()
    print("lead (GPU done - host done) per step, ms:", [round(ev0.elapsed_time(e) - 1e3 * h, 1) for e, h in evs])


if __name__ == "__main__":
    main()
