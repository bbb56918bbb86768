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
    ap.add_argument("--gc", default="default", choices=["default", "freeze", "disable"], help="freeze: gc.collect() + gc.freeze() after the warm-up steps; disable: gc.disable()")
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
    # host-to-device uploads (Tensor.to from a CPU tensor): from pageable memory they block the host until the stream has reached them
    up = [0, 0.0]
    orig_to = torch.Tensor.to

    def to_timed(self, *a, **k):
        if self.device.type != "cpu":
            return orig_to(self, *a, **k)
        t = time.perf_counter()
        try:
            return orig_to(self, *a, **k)
        finally:
            up[0] += 1; up[1] += time.perf_counter() - t
    torch.Tensor.to = to_timed
    for i in range(args.warmup):
        w.step(i, bench.baskets(pool, i, n_d))
    torch.cuda.synchronize()
    import gc
    gc_log = []

    def on_gc(phase, info, _t=[0.0]):
        if phase == "start":
            _t[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], time.perf_counter() - _t[0], info.get("collected", 0)))
    gc.callbacks.append(on_gc)
    if args.gc == "freeze":
        gc.collect(); gc.freeze()
    elif args.gc == "disable":
        gc.disable()
    print(f"gc mode {args.gc}: thresholds {gc.get_threshold()}, counts {gc.get_count()}, tracked objects {len(gc.get_objects())}")
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    evs = []
    seg_prev = torch.cuda.memory_stats().get("segment.all.allocated", 0)
    for i in range(args.steps):
        acc.clear()
        ts = time.perf_counter()
        w.step(args.warmup + i, bench.baskets(pool, args.warmup + i, n_d))
        host = time.perf_counter() - ts
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((e, time.perf_counter() - t0))
        inside = sum(v[1] for v in acc.values())
        top = sorted(acc.items(), key=lambda kv: -kv[1][1])[:4]
        print(f"step {i}: garbage collections " + (", ".join(f"gen{g} {t * 1e3:.1f} ms" for g, t, _ in gc_log) or "none"))
        gc_log.clear()
        print(f"step {i}: Tensor.to() from CPU tensors: {up[0]} calls, {up[1] * 1e3:.1f} ms")
        up[0], up[1] = 0, 0.0
        st = torch.cuda.memory_stats()
        seg = st.get("segment.all.allocated", 0)
        print(f"step {i}: new allocator segments (hipMalloc) {seg - seg_prev}, reserved {st.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB, peak allocated {st.get('allocated_bytes.all.peak', 0) / 2**30:.1f} GiB")
        seg_prev = seg
        print(f"step {i}: host {host * 1e3:7.1f} ms, inside C-ABI calls {inside * 1e3:7.1f} ms ({sum(v[0] for v in acc.values())} calls) | " +
              " | ".join(f"{k} n={v[0]} {v[1] * 1e3:.1f} ms max {v[2] * 1e3:.2f}" for k, v in top))
    torch.cuda.synchronize()
    print("lead (GPU done - host done) per step, ms:", [round(ev0.elapsed_time(e) - 1e3 * h, 1) for e, h in evs])


if __name__ == "__main__":
    main()
