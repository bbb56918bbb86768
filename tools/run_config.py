"""Train a reference configuration file for a few steps on synthetic data and print ONE JSON line (images / second of the G+D step, last losses):
the shortest path from `src/configs/<DATA>/<NAME>.yaml` to this package's training step (studiogan_amd.config_map.build -> worker.Worker.step).

  python tools/run_config.py <file.yaml> [--steps 20] [--warmup 3] [--bf16] [--batch N] [--set MODEL.g_conv_dim=8 ...] [--emulate]

--emulate runs the kernel sources on the CPU interpreter of tests/hipemu (test infrastructure: minutes per step at real widths; use --set to shrink the networks)
instead of a GPU; without it the script needs cuda:0 and libsgamd.so and fails loudly otherwise (the product path has no CPU fallback)."""
import argparse
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bf16", action="store_true", help="mixed precision (RUN.mixed_precision of the reference; bf16 here, scaler-free)")
    ap.add_argument("--batch", type=int, default=0, help="override OPTIMIZATION.batch_size")
    ap.add_argument("--set", action="append", default=[], metavar="SECTION.key=value", help="override one configuration entry (value parsed as YAML)")
    ap.add_argument("--emulate", action="store_true")
    a = ap.parse_args()
    y = yaml.safe_load(open(a.config))
    for item in a.set:
        path, val = item.split("=", 1)
        sec, key = path.split(".", 1)
        y.setdefault(sec, {})[key] = yaml.safe_load(val)
    if a.batch:
        y.setdefault("OPTIMIZATION", {})["batch_size"] = a.batch
    emu = None
    if a.emulate:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        import fullemu
        torch.set_num_threads(1)
        emu = fullemu.Installed(dma_late=1, greedy=1, seed=1)
        emu.__enter__()
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "run_config.py needs a GPU (or --emulate)"
        dev = torch.device("cuda", 0)
    import studiogan_amd  # noqa: F401
    from studiogan_amd import config_map as CM
    torch.manual_seed(1234)
    G, D, w = CM.build(y, dev, mixed_precision=a.bf16)
    kw = CM.worker_kwargs(y)
    B, S, nc = kw["batch_size"], (y.get("DATA") or {}).get("img_size", 32), kw["num_classes"]
    need = kw["d_updates_per_step"] * kw["acml_steps"]
    # synthetic real batches on the uint8 grid the reference's ToTensor + Normalize(0.5, 0.5) produces (src/data_util.py:92-94)
    pool = [((torch.randint(0, 256, (B, 3, S, S)).float() / 127.5 - 1.0).to(dev), torch.randint(0, nc, (B,)).to(dev)) for _ in range(max(need, 4))]

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def reals(i):
        return [pool[(i * need + k) % len(pool)] for k in range(need)]

    for i in range(a.warmup):
        w.step(i, reals(i))
    sync()
    t0 = time.perf_counter()
    last = None
    for i in range(a.steps):
        last = w.step(a.warmup + i, reals(a.warmup + i))
    sync()
    dt = time.perf_counter() - t0
    d, g = (float(last[0]), float(last[1])) if last is not None else (float("nan"), float("nan"))
    print(json.dumps({"config": os.path.basename(a.config), "backbone": CM.model_args(y)[0], "device": "interpreter" if a.emulate else torch.cuda.get_device_name(0),
                      "dtype": "bf16" if a.bf16 else "f32", "batch": B, "img_size": S, "steps": a.steps, "warmup": a.warmup,
                      "ms_per_step": round(1e3 * dt / max(a.steps, 1), 3), "img_per_sec": round(B * kw["acml_steps"] * a.steps / dt, 2) if a.steps else None,
                      "d_loss": d, "g_loss": g, "data": "synthetic"}))
    if emu is not None:
        emu.__exit__(None, None, None)


if __name__ == "__main__":
    main()
