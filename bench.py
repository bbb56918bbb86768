"""bench.py -- images/sec of one StudioGAN training step (n_d discriminator updates + 1 generator update incl. optimizer,
EMA and gradient exchange; reference src/loader.py:392-405) on the HIP path.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json metric): BigGAN ImageNet-128 (big_resnet, ch 96, cBN + projection D, SN on G and D, attention
G@64^2 / D@64^2, hinge, Adam(5e-5/2e-4, betas (0, .999), eps 1e-6), d_updates_per_step 2, EMA on), batch 256 per GPU,
bf16 compute (fp32 master weights / statistics / optimizer), synthetic ImageNet-shaped inputs resident in HBM.
Weak scaling: every rank runs batch 256; gradients are all-reduced over RCCL once per update, G's batch norms are
synchronised (one fp64 all-reduce per BN layer and direction).

Prints ONE JSON line (rank 0): value = global images per second; roofline = achieved algorithmic TFLOP/s of the dominant
kernel family (the MFMA implicit-GEMM convolution engine) measured live with hipEvents on the launch stream;
cpu_baseline = the CPU oracle (restatement of the reference, oracle/restate.py) timed on the host cores at a reduced batch.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic conv/matmul FLOPs per image of batch per step for C3: 2*D_upd + 1*G_upd
C3_GFLOP_PER_IMG_STEP = 514.79
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


class _MODEL:
    info_type = "N/A"


WORKLOADS = {
    # name: (img_size, ch, z_dim, shared_dim, num_classes, attn_g, attn_d, n_d, g_lr, d_lr, beta1, beta2, gflop_per_img_step)
    "biggan128": dict(img_size=128, ch=96, z_dim=120, shared=128, classes=1000, attn_g=[4], attn_d=[1], n_d=2, g_lr=5e-5, d_lr=2e-4,
                      beta1=0.0, beta2=0.999, gflop=C3_GFLOP_PER_IMG_STEP, desc="BigGAN ImageNet-128 cBN+PD hinge + self-attention, EMA on (C3)"),
    "biggan32": dict(img_size=32, ch=96, z_dim=80, shared=128, classes=10, attn_g=[2], attn_d=[1], n_d=2, g_lr=5e-5, d_lr=2e-4,
                     beta1=0.0, beta2=0.999, gflop=None, desc="BigGAN CIFAR-sized smoke workload (not a benchmark configuration)"),
}


# kernel families of the convolution engine as the in-library launch profiler tags them (csrc/common.h SG_ENG_*)
ENGINES = ["other", "sg_conv_sk_kernel", "sg_conv_rs_kernel / sg_conv_rs96_kernel", "sg_conv_v4_kernel", "sg_conv_v3_kernel", "sg_conv_v2_kernel",
           "sg_gemm_kernel<ConvPix>", "sg_conv_v4_kernel<SKIP>", "sg_conv_q_kernel", "sg_conv_q_kernel<SKIP>", "sg_wgrad_sk_kernel", "sg_wgrad_v3_kernel",
           "sg_wgrad_v2_kernel", "sg_gemm_kernel<wgrad>", "sg_wgrad_q_kernel"]


def peak_for(mixed):
    return PEAK_BF16_TFLOPS if mixed else PEAK_F32_TFLOPS


def csrc_sha16():
    """sha256 (first 16 hex digits) over csrc/*.{h,hip} -- the sources of every kernel the benchmarked configurations launch (the step, the extras, the
    FID leg): the PMC summaries under profiles/ carry the same figure (tools/pmc_traffic.py), so the line can say whether its byte counts were taken
    on THIS code (the GPU box has no .git to ask). csrc/ext/ (augmentations in front of the discriminator: no benchmarked configuration uses them) is not
    part of it."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pytorch-studiogan_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# rocprof names of the kernels behind a profiler tag (the lean weight-gradient kernels kept their tags)
TAG_KERNELS = {"sg_wgrad_q_kernel": ("sg_wgrad_q_kernel", "sg_wgrad_ql_kernel"), "sg_wgrad_v3_kernel": ("sg_wgrad_v3_kernel", "sg_wgrad_v3l_kernel"),
               "sg_conv_rs_kernel": ("sg_conv_rs_kernel", "sg_conv_rs96_kernel")}


def per_kernel_table(L, nsteps, peak, pmc=None):
    """roofline.per_kernel: the convolution engine's launches of the timed region by kernel family (hipEvent time per launch, recorded on the launch
    stream by libsgamd.so): launches / step, ms / step, algorithmic and executed TFLOP/s, fraction of the MFMA peak, algorithmic HBM bytes per launch,
    and -- when a PMC summary of the same command is committed under profiles/ -- measured HBM bytes / algorithmic bytes."""
    n = len(ENGINES)
    t = (ctypes.c_double * (n * 5))()
    L.call("sg_prof_collect_tags", t, n)
    rows = {}
    for i, name in enumerate(ENGINES):
        cnt, ms, fl, ex, by = t[i * 5:i * 5 + 5]
        if cnt == 0:
            continue
        r = {"launches_per_step": round(cnt / nsteps, 1), "ms_per_step": round(ms / nsteps, 3),
             "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None, "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 else None,
             "executed_tflops": round(ex / (ms * 1e-3) / 1e12, 1) if ms > 0 else None, "executed_frac": round(ex / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 else None,
             "algorithmic_MB_per_launch": round(by / cnt / 1e6, 1), "algorithmic_GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
        if pmc:
            # measured HBM bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, tools/pmc_traffic.py) of the family's
            # kernel templates over the algorithmic bytes; a weight-gradient family includes the split-K reduce launches that follow its kernels
            key = name.split(" ")[0].split("<")[0]
            skip = "SKIP" in name
            def targs(k):
                a = k[k.index("<") + 1:k.rindex(">")] if "<" in k and ">" in k else ""
                return [t.strip() for t in a.split(",")]

            def member(k):
                if not any(kk + "<" in k or k.endswith(kk) for kk in TAG_KERNELS.get(key, (key,))):
                    return False
                if key == "sg_conv_v4_kernel":      # template <NB, RELU, UP, TJW, SKIP>
                    a = targs(k)
                    return (len(a) > 4 and a[4] == "true") == skip
                if key == "sg_conv_q_kernel":       # template <NB, RELU, TJW, SKIP, NPMIN>
                    a = targs(k)
                    return (len(a) > 3 and a[3] == "true") == skip
                if key == "sg_gemm_kernel":
                    return "ConvPix" in k
                return True
            meas = [v for v in pmc if member(v["kernel"])]
            tot_b = sum(v["bytes_per_launch"] * v["launches"] for v in meas)
            tot_n = sum(v["launches"] for v in meas)
            if tot_n and by > 0:
                r["pmc_MB_per_launch"] = round(tot_b / tot_n / 1e6, 1)
                r["pmc_bytes_over_algorithmic"] = round((tot_b / tot_n) / (by / cnt), 3)
        rows[name] = r
    return rows


def build(wl, mixed, device):
    from studiogan_amd import ops
    from studiogan_amd.backbones import big_resnet
    MOD = ops.Modules(apply_g_sn=True, apply_d_sn=True, g_cond_mtd="cBN", backbone="big_resnet")
    G = big_resnet.Generator(wl["z_dim"], wl["shared"], wl["img_size"], wl["ch"], True, wl["attn_g"], "cBN", wl["classes"], "ortho", "N/A",
                             mixed, MOD, _MODEL).to(device)
    D = big_resnet.Discriminator(wl["img_size"], wl["ch"], True, True, wl["attn_d"], "PD", "W/O", "N/A", False, wl["classes"], "ortho", "N/A",
                                 mixed, MOD, _MODEL).to(device)
    return G, D


# ---- extra workloads (never the headline): the other BASELINE.json configurations, a few steps each ---------------------------------
EXTRAS = {
    # C1: configs/CIFAR10/GGAN.yaml (DCGAN backbone, hinge, BN in G and D, no SN), batch 64, fp32
    "dcgan32_bs64_fp32": dict(yaml={"DATA": {"img_size": 32, "num_classes": 10}, "MODEL": {"backbone": "deep_conv", "g_conv_dim": "N/A", "d_conv_dim": "N/A"}},
                              batch=64, mixed=False, n_d=2, loss="hinge", g_lr=2e-4, d_lr=2e-4, beta1=0.5, beta2=0.999, gp=False, ema=False,
                              desc="DCGAN CIFAR-10 32x32 unconditional hinge, batch 64, fp32 (C1)", steps=20, warmup=3),
    # C2: configs/CIFAR10/SNGAN.yaml at batch 256, fp32 (exact-fp32 MFMA: 157.3 TFLOP/s peak)
    "sngan32_bs256_fp32": dict(yaml={"DATA": {"img_size": 32, "num_classes": 10},
                                     "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "z_dim": 128, "g_conv_dim": 64, "d_conv_dim": 64}},
                               batch=256, mixed=False, n_d=5, loss="hinge", g_lr=2e-4, d_lr=2e-4, beta1=0.5, beta2=0.999, gp=False, ema=False,
                               desc="SNGAN CIFAR-10 32x32 cBN+PD hinge, batch 256, fp32 (C2)"),
    # C2 once more with the generic engine's fp32 convolutions on the bf16x3 split-precision path (fp32 tensors, three bf16 MFMAs per k-tile on operands split in
    # registers: functional.f32_mode; forward, data gradient and weight gradient). NOT the exact-fp32 figure above: its own line, its own label.
    "sngan32_bs256_fp32_bf16x3": dict(yaml={"DATA": {"img_size": 32, "num_classes": 10},
                                            "MODEL": {"backbone": "resnet", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_d_sn": True, "z_dim": 128, "g_conv_dim": 64, "d_conv_dim": 64}},
                                      batch=256, mixed=False, n_d=5, loss="hinge", g_lr=2e-4, d_lr=2e-4, beta1=0.5, beta2=0.999, gp=False, ema=False, f32_mode="bf16x3",
                                      desc="SNGAN CIFAR-10 32x32 cBN+PD hinge, batch 256, fp32 tensors with bf16x3 split-precision convolution arithmetic (C2 variant)"),
    # C5: configs/CIFAR10/WGAN-GP.yaml at img_size 128 (ResNet, unconditional, BN in D, wasserstein + gradient penalty: double backward), bf16
    "wgangp128_bs64_bf16": dict(yaml={"DATA": {"img_size": 128, "num_classes": 1000},
                                      "MODEL": {"backbone": "resnet", "z_dim": 128, "g_conv_dim": 64, "d_conv_dim": 64}},
                                batch=64, mixed=True, n_d=5, loss="wasserstein", g_lr=2e-4, d_lr=2e-4, beta1=0.5, beta2=0.999, gp=True, ema=False,
                                desc="WGAN-GP ResNetGAN ImageNet-128 (gradient-penalty double backward), per-GPU batch 64, bf16 (C5)"),
    # C4 per GPU: configs/ImageNet/BigGAN-Deep-2048.yaml model at 128^2, per-GPU batch 256 (= 2048 / 8), bf16, EMA on
    "bigdeep128_bs256_bf16": dict(yaml={"DATA": {"img_size": 128, "num_classes": 1000},
                                        "MODEL": {"backbone": "big_resnet_deep_legacy", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                                                  "apply_attn": True, "attn_g_loc": [4], "attn_d_loc": [1], "z_dim": 128, "g_shared_dim": 128, "g_conv_dim": 128,
                                                  "d_conv_dim": 128, "g_depth": 2, "d_depth": 2}},
                                  batch=256, mixed=True, n_d=2, loss="hinge", g_lr=5e-5, d_lr=2e-4, beta1=0.0, beta2=0.999, gp=False, ema=True,
                                  desc="BigGAN-Deep ImageNet-128 ch 128 depth 2, per-GPU batch 256 of the 2048 global batch, bf16, EMA on (C4 per GPU)"),
    # C4 at the resolution BASELINE.json names: the same model at img_size 256 (reference src/models/big_resnet_deep_legacy.py:80-95: the "256"
    # tables; attention in D at 128^2 = 16384 positions), 64 per GPU per micro-step (2048 = 8 GPUs x 64 x 4 accumulation steps)
    "bigdeep256_bs64_bf16": dict(yaml={"DATA": {"img_size": 256, "num_classes": 1000},
                                       "MODEL": {"backbone": "big_resnet_deep_legacy", "g_cond_mtd": "cBN", "d_cond_mtd": "PD", "apply_g_sn": True, "apply_d_sn": True,
                                                 "apply_attn": True, "attn_g_loc": [4], "attn_d_loc": [1], "z_dim": 128, "g_shared_dim": 128, "g_conv_dim": 128,
                                                 "d_conv_dim": 128, "g_depth": 2, "d_depth": 2}},
                                 batch=64, mixed=True, n_d=2, loss="hinge", g_lr=5e-5, d_lr=2e-4, beta1=0.0, beta2=0.999, gp=False, ema=True,
                                 desc="BigGAN-Deep ImageNet-256 ch 128 depth 2, per-GPU micro-batch 64 (2048 = 8 x 64 x 4 accumulation steps), bf16, EMA on (C4 at 256^2)"),
}


def build_from_cfg(y, mixed, device):
    import importlib
    from studiogan_amd import ops
    M, D = y["MODEL"], y["DATA"]
    bb = importlib.import_module("studiogan_amd.backbones." + M.get("backbone", "resnet"))
    MOD = ops.Modules(apply_g_sn=M.get("apply_g_sn", False), apply_d_sn=M.get("apply_d_sn", False), g_cond_mtd=M.get("g_cond_mtd", "W/O"),
                      backbone=M.get("backbone", "resnet"))
    G = bb.Generator(M.get("z_dim", 128), M.get("g_shared_dim", "N/A"), D["img_size"], M.get("g_conv_dim", 64), M.get("apply_attn", False),
                     M.get("attn_g_loc", ["N/A"]), M.get("g_cond_mtd", "W/O"), D["num_classes"], "ortho", M.get("g_depth", "N/A"), mixed, MOD, _MODEL)
    Dm = bb.Discriminator(D["img_size"], M.get("d_conv_dim", 64), M.get("apply_d_sn", False), M.get("apply_attn", False), M.get("attn_d_loc", ["N/A"]),
                          M.get("d_cond_mtd", "W/O"), "W/O", "N/A", False, D["num_classes"], "ortho", M.get("d_depth", "N/A"), mixed, MOD, _MODEL)
    return G.to(device), Dm.to(device)


def run_extra(name, device, steps=None, warmup=None):
    """A few timed steps of one extra workload on ONE GPU: images/s, conv-engine TFLOP/s (hipEvent brackets) against the peak of its dtype."""
    import math
    from studiogan_amd import _lib as L
    from studiogan_amd.worker import Worker
    e = EXTRAS[name]
    y = e["yaml"]
    steps, warmup = steps or e.get("steps", 2), warmup or e.get("warmup", 1)
    torch.manual_seed(4321)
    G, D = build_from_cfg(y, e["mixed"], device)
    w = Worker(G, D, y["MODEL"].get("z_dim", 128), y["DATA"]["num_classes"], e["batch"], e["loss"], e["g_lr"], e["d_lr"], e["beta1"], e["beta2"],
               d_updates_per_step=e["n_d"], apply_g_ema=e["ema"], g_ema_decay=0.9999, g_ema_start=20000, apply_gp=e["gp"], gp_lambda=10.0)
    pool = generator_real_pool(G, min(32, e["n_d"] * (warmup + steps)), e["batch"], y["MODEL"].get("z_dim", 128), y["DATA"]["num_classes"], device, 77)
    L.call("sg_set_f32_mode", 3 if e.get("f32_mode") == "bf16x3" else 0)
    try:
        for i in range(warmup):
            w.step(i, baskets(pool, i, e["n_d"]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = w.step(warmup + i, baskets(pool, warmup + i, e["n_d"]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the convolution engine's share: event pairs around every engine launch of ONE further step (inside the timed steps the pairs -- two barrier packets per launch,
        # thousands per step on the small-batch configurations -- would slow what is timed: profiles/r06_bench_instrumentation_ab_r7e.txt)
        psteps = 1
        L.call("sg_prof_enable", 1)
        for i in range(psteps):
            w.step(warmup + steps + i, baskets(pool, warmup + steps + i, e["n_d"]))
        torch.cuda.synchronize()
        pr = (ctypes.c_double * 9)()
        L.call("sg_prof_collect", pr, 3)
        L.call("sg_prof_enable", 0)
    finally:
        L.call("sg_set_f32_mode", 0)
    d_l, g_l = float(last[0]), float(last[1])
    assert math.isfinite(d_l) and math.isfinite(g_l), f"{name}: non-finite losses D {d_l} G {g_l}"
    conv_ms, conv_fl = pr[1] + pr[4], pr[2] + pr[5]
    peak = PEAK_BF16_TFLOPS if e["mixed"] else PEAK_F32_TFLOPS
    if e.get("f32_mode") == "bf16x3":
        peak = round(PEAK_BF16_TFLOPS / 3.0, 1)      # three bf16 MFMAs per fp32-equivalent MAC: the matrix pipe's ceiling for this arithmetic
    tf = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    if e["loss"] == "hinge":
        assert d_l > 1e-2, f"{name}: the discriminator saturated (d_loss {d_l}): its backward would multiply zero gradients"
    del w, G, D, pool
    torch.cuda.empty_cache()
    return {"workload": e["desc"], "dtype": "bf16" if e["mixed"] else ("f32 tensors, bf16x3 arithmetic" if e.get("f32_mode") == "bf16x3" else "f32"), "per_gpu_batch": e["batch"], "d_updates_per_step": e["n_d"], "steps": steps,
            "images_per_sec": round(e["batch"] * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 2),
            "conv_engine_tflops": round(tf, 1), "conv_engine_frac_of_peak": round(tf / peak, 4), "peak_tflops": peak,
            "conv_ms_per_step": round(conv_ms / psteps, 2), "step_conv_gflop_per_image": round(conv_fl / psteps / e["batch"] / 1e9, 2),
            "conv_measured_on": f"{psteps} profiled step after the {steps} timed ones",
            "last_step_losses": {"d_loss": round(d_l, 5), "g_loss": round(g_l, 5)}}


def synth_batches(n, batch, img_size, classes, device, seed):
    """uint8-grid 'real' images in [-1,1] and labels, generated on the CPU with a seeded generator."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        k = torch.randint(0, 256, (batch, 3, img_size, img_size), generator=g, dtype=torch.int32)
        x = (k.float() / 127.5 - 1.0).to(device)
        y = torch.randint(0, classes, (batch,), generator=g).to(device)
        out.append((x, y))
    return out


def generator_real_pool(G, n_baskets, batch, z_dim, classes, device, seed):
    """Synthetic 'real' baskets that keep the discriminator's task HARD, so the hinge stays active for the whole timed region:
    uint8-quantised samples of a frozen copy of the INITIAL generator (train-mode batch statistics, no tracking), labels random.
    Uniform-noise 'real' images are separated from generator samples with margin > 1 within ~10 updates; from then on both hinge
    terms are exactly zero, dy into the discriminator's backward is identically zero (34 % of the step's FLOPs multiply zeros) and
    the chip clocks higher on the zero operands (MI355X_MICROARCH.md, DVFS give-back) -- VERDICT r2 weak item 3. The reference's
    loop draws a fresh basket per D update (src/worker.py:227-304); here n_baskets distinct baskets are cycled."""
    import copy
    from studiogan_amd.worker import untrack_bn_statistics
    G0 = copy.deepcopy(G)
    G0.train()
    G0.apply(untrack_bn_statistics)
    g = torch.Generator(device=device).manual_seed(seed)
    out = []
    with torch.no_grad():
        for _ in range(n_baskets):
            z = torch.randn(batch, z_dim, device=device, generator=g)
            y = torch.randint(0, max(classes, 1), (batch,), device=device, generator=g)
            x = G0(z, y).float().clamp_(-1.0, 1.0)
            x = torch.round((x + 1.0) * 127.5) / 127.5 - 1.0          # the uint8 grid of ToTensor + Normalize(0.5, 0.5)
            yr = torch.randint(0, max(classes, 1), (batch,), device=device, generator=g)
            out.append((x.contiguous(), yr))
    del G0
    torch.cuda.empty_cache()
    return out


def baskets(pool, step, n_d):
    """the n_d real micro-batches of training step `step` out of a cycled pool"""
    return [pool[(step * n_d + k) % len(pool)] for k in range(n_d)]


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a container can see 100+ host
    cores while being entitled to a handful -- sizing the torch thread pool by os.cpu_count() then thrashes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline_child(workload, cpu_batch):
    """Runs in a subprocess (hard wall-clock bound enforced by the parent). The reference path restated on CPU
    (oracle/restate.py / oracle/inception.py), fp32, same architecture and step structure. Timed only -- never used to
    produce the GPU result. workload: "biggan128" (C3, reduced batch), a key of EXTRAS (C1 / C2 at their FULL batch when cpu_batch == 0),
    or "inception" (InceptionV3 forward at 299^2, SURVEY.md 8d: B = 32)."""
    threads = usable_cores()
    torch.set_num_threads(threads)
    if workload == "inception":
        from oracle import inception as OI
        sd = OI.random_state_dict(0)
        B = cpu_batch or 32
        x = torch.rand(B, 3, 299, 299, generator=torch.Generator().manual_seed(5)) * 2 - 1
        with torch.no_grad():
            t0 = time.time()
            OI.inception_forward(x, sd)
            first = time.time() - t0
            times = []
            while len(times) < 3 and (not times or sum(times) + times[-1] < 25.0):
                t0 = time.time()
                OI.inception_forward(x, sd)
                times.append(time.time() - t0)
        dt = sum(times) / len(times)
        print(json.dumps({"value": round(B / dt, 3), "unit": "samples/sec", "cores": threads, "kind": "port",
                          "sample": f"{len(times)} timed InceptionV3 forward pass(es) after 1 warm-up at batch {B}, 299x299 fp32 inputs already resized, torch CPU ops "
                                    f"through oracle/inception.py (restatement of reference src/metrics/inception_net.py over torchvision's structure): {dt:.2f} s/batch"}))
        return
    from oracle import restate as O
    if workload in WORKLOADS:
        wl = WORKLOADS[workload]
        ocfg = dict(img_size=wl["img_size"], g_conv_dim=wl["ch"], d_conv_dim=wl["ch"], z_dim=wl["z_dim"], attn_g_loc=wl["attn_g"],
                    attn_d_loc=wl["attn_d"], apply_attn=True, g_cond_mtd="cBN", d_cond_mtd="PD", apply_d_sn=True, backbone="big_resnet")
        Gc, Dc = build(wl, False, torch.device("cpu"))
        img, z_dim, classes, n_d, loss = wl["img_size"], wl["z_dim"], wl["classes"], wl["n_d"], "hinge"
        lrs = (wl["g_lr"], wl["d_lr"], wl["beta1"], wl["beta2"])
        desc = "BigGAN-128 ch 96, n_d = 2"
    else:
        from oracle.make_golden import oracle_cfg
        e = EXTRAS[workload]
        y = e["yaml"]
        ocfg = oracle_cfg(y)
        Gc, Dc = build_from_cfg(y, False, torch.device("cpu"))
        img, z_dim, classes, n_d, loss = y["DATA"]["img_size"], y["MODEL"].get("z_dim", 128), y["DATA"]["num_classes"], e["n_d"], e["loss"]
        lrs = (e["g_lr"], e["d_lr"], e["beta1"], e["beta2"])
        cpu_batch = cpu_batch or e["batch"]
        desc = e["desc"]
        assert not e["gp"], "gradient-penalty CPU baseline not wired"
    isb = lambda k: any(s in k for s in ("weight_u", "weight_v", "running_", "num_batches"))
    GP = {k: v.detach().clone() for k, v in Gc.state_dict().items() if not isb(k)}
    GB = {k: v.detach().clone() for k, v in Gc.state_dict().items() if isb(k)}
    DP = {k: v.detach().clone() for k, v in Dc.state_dict().items() if not isb(k)}
    DB = {k: v.detach().clone() for k, v in Dc.state_dict().items() if isb(k)}
    del Gc, Dc
    gen_fn, dis_fn = O.model_fns(ocfg)
    g_opt, d_opt = O.AdamState(GP, lrs[0], lrs[2], lrs[3]), O.AdamState(DP, lrs[1], lrs[2], lrs[3])
    gen = torch.Generator().manual_seed(99)

    def one_step():
        for _ in range(n_d):
            z = torch.randn(cpu_batch, z_dim, generator=gen)
            fl = torch.randint(0, classes, (cpu_batch,), generator=gen)
            real = torch.randint(0, 256, (cpu_batch, 3, img, img), generator=gen).float() / 127.5 - 1.0
            rl = torch.randint(0, classes, (cpu_batch,), generator=gen)
            O.d_update(gen_fn, dis_fn, GP, GB, DP, DB, d_opt, [real], [rl], [z], [fl], loss)
        z = torch.randn(cpu_batch, z_dim, generator=gen)
        fl = torch.randint(0, classes, (cpu_batch,), generator=gen)
        O.g_update(gen_fn, dis_fn, GP, GB, DP, DB, g_opt, [z], [fl], loss)
    t0 = time.time()
    one_step()
    first = time.time() - t0
    sys.stderr.write(f"[cpu_baseline] {workload}: warm-up step {first:.1f}s on {threads} threads\n")
    # 1 warm-up + 3 timed steps; bounded: stop timing once ~40 s of timed work are spent (at least one timed step)
    times = []
    while len(times) < 3 and (not times or sum(times) + times[-1] < 40.0):
        t0 = time.time()
        one_step()
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    print(json.dumps({"value": round(cpu_batch / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
                      "sample": f"{len(times)} timed G+D step(s) after 1 warm-up at batch {cpu_batch} of the same architecture ({desc}), fp32, "
                                f"torch CPU ops through oracle/restate.py (restatement of the reference, bit-identical to it on the golden fixtures): "
                                f"{dt:.2f} s/step (warm-up {first:.2f} s)"}))


def cpu_baseline(workload, cpu_batch, timeout_s=170, unit="images/sec"):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--workload", workload, "--cpu-batch", str(cpu_batch)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
        line = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
        if line:
            return json.loads(line[-1])
        return {"value": None, "unit": unit, "cores": usable_cores(), "kind": "port", "sample": "cpu baseline failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": unit, "cores": usable_cores(), "kind": "port",
                "sample": f"the {workload} CPU sample at batch {cpu_batch} did not finish within the {timeout_s}s bound"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the first steps after process start run 2-3 % slower on the GPU side (r7q, same box: 5 steps after 2 warm-up 128.3 ms, 20 after 5 124.85 ms, the host > 100 ms ahead in
    # both): ten timed steps after five warm-up steps are representative and still take < 2 s
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="biggan128")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--fp32", action="store_true", help="fp32 compute instead of bf16 (not the benchmark configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fid-samples", type=int, default=50000, help="samples of the FID feature-extraction leg (0 = skip)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads (C2 fp32, C5 WGAN-GP, C4 per GPU)")
    ap.add_argument("--native-comm", action="store_true", help="same as --comm native")
    ap.add_argument("--comm", choices=["auto", "native", "torch"], default=os.environ.get("SG_COMM", "auto"),
                    help="N > 1: who carries the exchanges. auto (default): libsgamd.so's own RCCL communicators for the gradient arena (sg_allreduce_flat on HIP "
                         "streams) and the peer-store mailboxes for sync-BN (exchange fused into the statistics kernel), each kept only if its self-test agrees with "
                         "torch.distributed on EVERY rank -- otherwise that exchange falls back to torch.distributed's nccl backend and the line says so; native: the "
                         "same without the fallback (a failing self-test aborts); torch: torch.distributed only")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --batch is the GLOBAL batch, every rank takes batch / world of it (what the reference does: "
                    "src/loader.py:162 divides the configured batch over the ranks); default is weak scaling (--batch per GPU)")
    ap.add_argument("--strict", action="store_true", help="exit non-zero when an extra workload / leg fails (the line is still printed)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child(args.workload, args.cpu_batch)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher of N ranks, one per GPU (what mp.spawn does for the reference,
        # src/main.py:178-188). The ranks run this same file under torch.distributed.run and rank 0 prints the JSON line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("[bench] launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # SG_BENCH_ONE_DEVICE=1: plumbing check of the multi-rank path on a 1-GPU box (all ranks on cuda:0, gloo instead of RCCL);
    # never a benchmark configuration
    one_dev = os.environ.get("SG_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_dev:
            dist.init_process_group(backend="gloo", init_method="env://")
        else:
            dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
        group = dist.group.WORLD
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different job size")
    if args.strong:
        if args.batch % world:
            raise SystemExit(f"bench.py --strong: global batch {args.batch} is not a multiple of the {world} ranks")
        args.batch //= world            # from here on args.batch is the per-rank batch, as in the weak-scaling run

    import studiogan_amd
    from studiogan_amd import _lib as L
    from studiogan_amd import ops
    from studiogan_amd.worker import Worker

    rccl_ranks, bn_exchange, comm_notes = None, None, []
    if args.native_comm or os.environ.get("SG_NATIVE_COMM") == "1":
        args.comm = "native"
    if world > 1 and args.comm != "torch":
        from studiogan_amd import comm as sg_comm

        def agreed(ok):          # a path is used only if EVERY rank saw it pass (one MIN all-reduce through torch.distributed)
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if not one_dev else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        want = world * (world + 1) / 2.0
        # 1. gradient arena: the C ABI's own RCCL communicators (RCCL refuses two ranks on one device: not on the one-device plumbing run)
        if not one_dev:
            ok, why = False, ""
            try:
                nc = sg_comm.enable(group, device=device)
                n = ctypes.c_int(0)
                L.call("sg_comm_size", nc.handle, ctypes.byref(n))
                t = torch.full((4096,), float(rank + 1), device=device)
                nc.allreduce_(t)
                t2 = torch.full((4096,), float(rank + 1), device=device)
                nc.stream.wait_stream(torch.cuda.current_stream())
                nc.allreduce_(t2, stream=nc.stream.cuda_stream, grad=True)          # (the gradient communicator, on its side stream)
                torch.cuda.synchronize()
                ok = n.value == world and bool((t == want).all()) and bool((t2 == want).all())
                why = "" if ok else f"self-test mismatch (ranks {n.value}, sum {float(t[0])} / {float(t2[0])}, expected {want})"
            except Exception as e:      # noqa: BLE001
                why = f"{type(e).__name__}: {str(e)[:200]}"
            if agreed(ok):
                rccl_ranks = world
            else:
                comm_notes.append("native RCCL communicator rejected: " + (why or "another rank failed its self-test"))
                sg_comm._REGISTRY.clear()
                if args.comm == "native":
                    raise SystemExit("bench.py --comm native: " + comm_notes[-1])
        # 2. sync-BN: peer-store mailboxes, the exchange fused into the finalize kernel (csrc/p2p.hip)
        ok, why = False, ""
        try:
            box = sg_comm.enable_p2p(group)
            v = torch.full((64,), float(rank + 1), dtype=torch.float64, device=device)
            for _ in range(3):
                v.fill_(float(rank + 1))
                box.allreduce_f64_(v)
            torch.cuda.synchronize()
            ok = bool((v == want).all()) and box.timeouts() == 0
            why = "" if ok else f"self-test mismatch (sum {float(v[0])}, expected {want}, timeouts {box.timeouts()})"
        except Exception as e:      # noqa: BLE001
            why = f"{type(e).__name__}: {str(e)[:200]}"
        if agreed(ok):
            bn_exchange = "peer-store mailboxes, exchange fused into the statistics kernel (sg_bn_finalize_p2p / sg_p2p_allreduce_f64)"
        else:
            comm_notes.append("peer-store mailboxes rejected: " + (why or "another rank failed its self-test"))
            sg_comm._P2P.clear()
            if args.comm == "native":
                raise SystemExit("bench.py --comm native: " + comm_notes[-1])
        if rank == 0:
            for note in comm_notes:
                sys.stderr.write("[bench] " + note + "\n")
    if world > 1 and bn_exchange is None:
        bn_exchange = "RCCL all-reduce between the statistics kernels (sg_bn_stats_sync)" if rccl_ranks else "torch.distributed all-reduce between the statistics kernels"
    wl = WORKLOADS[args.workload]
    mixed = not args.fp32
    torch.manual_seed(1234)  # identical initial weights on every rank (what DDP's initial broadcast guarantees)
    G, D = build(wl, mixed, device)
    if world > 1:
        for m in G.modules():
            if isinstance(m, ops.BatchNorm2d):
                m.sync_group = True
    w = Worker(G, D, wl["z_dim"], wl["classes"], args.batch, "hinge", wl["g_lr"], wl["d_lr"], wl["beta1"], wl["beta2"],
               d_updates_per_step=wl["n_d"], apply_g_ema=True, g_ema_decay=0.9999, g_ema_start=20000, group=group)
    torch.manual_seed(1234 + rank)  # per-rank sampling streams (reference src/loader.py:99)
    n_d = wl["n_d"]
    pool = generator_real_pool(G, min(32, n_d * (args.warmup + args.steps + 2)), args.batch, wl["z_dim"], wl["classes"], device, 1234 + rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tw = time.perf_counter()
    for i in range(args.warmup):
        w.step(i, baskets(pool, i, n_d))
    barrier()
    if rank == 0:
        sys.stderr.write(f"[bench] warmup {args.warmup} step(s): {time.perf_counter() - tw:.2f}s\n")
    # Event pairs in the TIMED region: around every launch of the dominant kernel (sg_conv2d_q, bit 2 of sg_prof_enable) and nothing else. Pairs around every engine launch
    # (~360 per step, the arrangement until round 6) are ~720 barrier packets per step: a traced run showed 5-20 us of idle in front of every bracketed kernel, ~3 ms per step
    # (tools/kt_gaps.py, profiles/r06_step_gaps_*.txt) -- the instrument slowed what it measured. The engine's other families, the family total and the HBM-bound families
    # are measured the same way on `psteps` further steps AFTER the timed region. SG_BENCH_PROF_ALL=1 restores the old arrangement (same-box A/B).
    prof_all = os.environ.get("SG_BENCH_PROF_ALL") == "1"
    L.call("sg_prof_enable", 1 if prof_all else 4)
    if world > 1:
        from studiogan_amd import comm as sg_comm_t
        sg_comm_t.timing(True)          # event pairs on the compute stream around every wait on a collective (comm.exposed)
    # host lead: how far ahead of the GPU the launching thread runs. One event per step (recorded on the compute stream, no sync) + the host clock at the moment the step's
    # launches have all been issued; lead_i = (GPU finish of step i) - (host finished issuing step i). A lead near zero = the GPU waits for the host in that step.
    ev0 = torch.cuda.Event(enable_timing=True)
    ev_steps, host_done = [], []
    t0 = time.perf_counter()
    ev0.record()
    last = None
    for i in range(args.steps):
        last = w.step(args.warmup + i, baskets(pool, args.warmup + i, n_d))
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev_steps.append(e)
        host_done.append(time.perf_counter() - t0)
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_done = [ev0.elapsed_time(e) for e in ev_steps]
    host_lead = {"host_issue_ms_per_step": round(1e3 * host_done[-1] / args.steps, 3),
                 "gpu_minus_host_ms_per_step_end": [round(g - 1e3 * h, 2) for g, h in zip(gpu_done, host_done)][:32],
                 "note": "per timed step: GPU completion time of the step minus the host time at which its last launch was issued (both from the start of the timed region); "
                         "positive and growing = the host runs ahead and the step is GPU-bound"}
    exposed_comm_ms = None
    if world > 1:
        exposed_comm_ms = sg_comm_t.timing_ms()
        sg_comm_t.timing(False)
        # a peer-store exchange whose partner never arrived trips a bounded spin and poisons the statistics instead of hanging: must not have happened
        box_ = sg_comm_t.p2p_for(group)
        p2p_timeouts = box_.timeouts() if box_ is not None else 0
        tx = torch.tensor([exposed_comm_ms, float(p2p_timeouts)], dtype=torch.float64, device=device)
        dist.all_reduce(tx, op=dist.ReduceOp.MAX)
        exposed_comm_ms, p2p_timeouts = float(tx[0].item()), int(tx[1].item())
        assert p2p_timeouts == 0, f"{p2p_timeouts} peer-store granule waits ran into their spin limit during the timed steps"
    # the step must have produced numbers: finite losses of the last timed step (read AFTER the timed region: a host sync)
    d_last, g_last = (float(last[0]), float(last[1])) if last is not None else (float("nan"), float("nan"))
    import math
    assert math.isfinite(d_last) and math.isfinite(g_last), f"non-finite losses after the timed steps: D {d_last} G {g_last}"
    # the hinge must still be active on the LAST timed step: a saturated discriminator (d_loss == 0) backpropagates exact zeros
    assert d_last > 1e-2, f"discriminator saturated in the timed region (d_loss {d_last}): the measured step would multiply zero gradients"
    fake_chk = w.last_g[0]
    assert bool(torch.isfinite(fake_chk).all()) and float(fake_chk.abs().max()) <= 1.0, "generator images of the last step are not finite / not in [-1, 1]"
    # PMC byte counts need their own rocprofv3 passes: the committed summary of those passes over this same command (tools/pmc_traffic.py). The
    # summary carries the hash of the kernel sources it was taken on; a summary of OTHER sources is reported as stale, never silently used as current
    pmc_tab, pmc_src, pmc_sha, cur_sha = None, None, None, csrc_sha16()
    try:
        pmc_name = next(n for n in ("r06_conv_hbm_traffic_pmc.json", "r05_conv_hbm_traffic_pmc.json", "r04_conv_hbm_traffic_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        pj = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        pmc_tab, pmc_sha = pj.get("per_kernel"), pj.get("csrc_sha16")
        pmc_src = "profiles/" + pmc_name
    except Exception:
        pass
    per_kernel_timed = per_kernel_table(L, args.steps, PEAK_BF16_TFLOPS if mixed else PEAK_F32_TFLOPS, pmc_tab)      # (prof_all: every family; else the dominant kernel's rows)
    prof4 = (ctypes.c_double * 12)()
    L.call("sg_prof_collect_ex", prof4, 3)
    L.call("sg_prof_enable", 0)
    per_kernel, psteps = per_kernel_timed, args.steps          # (replaced below by the profiled steps' table unless prof_all)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- north-star target figure: the BigGAN-128 (BigGAN-256.yaml) D FORWARD conv stack at batch 256 -----------
    # 21.673 GFLOP/img whole forward, 21.170 GFLOP/img conv-only; conv-only time = hipEvent brackets of the conv launches.
    dfwd = None
    if args.workload == "biggan128" and rank == 0:
        x, y = pool[0]
        with torch.no_grad():
            for _ in range(2):
                D(x, y)
            torch.cuda.synchronize()
            L.call("sg_prof_enable", 1)
            it = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                D(x, y)
            e1.record()
            torch.cuda.synchronize()
        dfwd_kernels = per_kernel_table(L, it, PEAK_BF16_TFLOPS if mixed else PEAK_F32_TFLOPS)
        pd4 = (ctypes.c_double * 12)()
        L.call("sg_prof_collect_ex", pd4, 3)
        L.call("sg_prof_enable", 0)
        pd = [pd4[(i // 3) * 4 + (i % 3)] for i in range(9)]
        fwd_ms = e0.elapsed_time(e1) / it
        conv_only_ms = pd[1] / it
        pk = PEAK_BF16_TFLOPS if mixed else PEAK_F32_TFLOPS
        alg_tf = pd[2] / it / (conv_only_ms * 1e-3) / 1e12 if conv_only_ms > 0 else None
        exe_tf = pd4[3] / it / (conv_only_ms * 1e-3) / 1e12 if conv_only_ms > 0 else None
        dfwd = {"what": "BigGAN-128 D forward at batch %d (SN power iteration + weight packing included in forward_ms)" % args.batch,
                # algorithmic = the reference's op graph (3x3 convolution + AvgPool2d as written in src/models/big_resnet.py:177-242) = SURVEY 8(d) / the north-star
                # target's accounting; executed = the MFMAs issued: the pooled block tails run as 4x4 / stride-2 convolutions with the pre-summed filter (csrc/conv_q.h)
                "conv_stack_frac_of_peak": round(alg_tf / pk, 4) if alg_tf else None,
                "conv_stack_executed_frac_of_peak": round(exe_tf / pk, 4) if exe_tf else None,
                "conv_stack_ms": round(conv_only_ms, 3), "conv_launches": int(pd[0] / it),
                "conv_stack_tflops": round(alg_tf, 1) if alg_tf else None, "conv_stack_executed_tflops": round(exe_tf, 1) if exe_tf else None, "peak_tflops": pk,
                "forward_ms": round(fwd_ms, 3), "forward_tflops": round(21.673 * args.batch / fwd_ms, 1),
                "per_kernel": dfwd_kernels}
    # ---- HBM-bound kernel families of the step: algorithmic bytes / hipEvent time per family over two more steps --------
    hbm = None
    if rank == 0 or world > 1:
        L.call("sg_prof_enable", 2 if prof_all else 3)          # bit 1: spectral norm, batch norm, attention score kernels, Adam / EMA; bit 0: every engine launch
        hsteps = 2
        for i in range(hsteps):
            w.step(args.warmup + args.steps + i, baskets(pool, args.warmup + args.steps + i, n_d))
        barrier()
        if not prof_all:
            per_kernel, psteps = per_kernel_table(L, hsteps, PEAK_BF16_TFLOPS if mixed else PEAK_F32_TFLOPS, pmc_tab), hsteps
        ph4 = (ctypes.c_double * 28)()
        L.call("sg_prof_collect_ex", ph4, 7)
        L.call("sg_prof_enable", 0)
        ph = [ph4[(i // 3) * 4 + (i % 3)] for i in range(21)]          # {launches, ms, algorithmic work} per kind
        if not prof_all:
            prof4 = ph4          # kinds 0-2 of the profiled steps
        if rank == 0:
            hbm = {"peak_GBps": 8000.0, "steps": hsteps, "note": "algorithmic bytes (each tensor of the family read / written once per pass) / hipEvent time on the launch stream"}
            for kind, name in ((3, "spectral_norm"), (4, "batch_norm"), (5, "attention_scores"), (6, "adam_ema")):
                n_l, ms, by = ph[kind * 3], ph[kind * 3 + 1], ph[kind * 3 + 2]
                gbps = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                hbm[name] = {"ms_per_step": round(ms / hsteps, 3), "GB_per_step": round(by / hsteps / 1e9, 3), "GBps": round(gbps, 1),
                             "frac_of_8TBps": round(gbps / 8000.0, 3), "calls_per_step": round(n_l / hsteps, 1)}
            if args.workload == "biggan128" and hbm["attention_scores"]["ms_per_step"] > 0:
                # The attention core is NOT an HBM family (scores / probabilities never leave the chip): its roofs are the matrix pipe and v_exp_f32. Both attention layers of
                # C3 sit at 64 x 64 with 192 channels (reference src/utils/ops.py:83-103: theta / phi 24 wide, g 96 wide, keys max-pooled to 1024). Per image and forward:
                # QK^T + PV = 2 * 4096 * 1024 * (24 + 96); the fused backward recomputes the scores in both of its kernels: q side QK^T + dO g^T + dS phi, k side the same two
                # products + dS^T theta + P^T dO. Per step: (n_d + 1) G + (2 n_d + 1) D forwards, 1 G + (2 n_d + 1) D backwards (a D update's backward crosses the attention of its real and its fake forward). One exponential per score per kernel that forms P.
                hw, keys, dqk, dv = 4096, 1024, 24, 96
                fwd = 2.0 * hw * keys * (dqk + dv)
                bwd = 2.0 * hw * keys * ((dqk + dv + dqk) + (dqk + dv + dqk + dv))
                n_f, n_b = 3 * n_d + 2, 2 * n_d + 2
                flop = args.batch * (n_f * fwd + n_b * bwd)
                exps = args.batch * hw * keys * (n_f + 2 * n_b)
                a_ms = hbm["attention_scores"]["ms_per_step"]
                hbm["attention_scores"]["compute_roof"] = {
                    "gflop_per_step": round(flop / 1e9, 1), "tflops": round(flop / (a_ms * 1e-3) / 1e12, 1), "frac_of_mfma_peak": round(flop / (a_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "exp_per_step": exps, "exp_bound_ms": round(exps / 9.8e12 * 1e3, 3),
                    "note": "unpadded dims (the kernels pad theta / phi from 24 to 32 channels); v_exp_f32 at 1/4 rate = 9.8e12 per second chip-wide; frac_of_8TBps above is kept "
                            "only to show that HBM is not the roof"}
    # ---- second metric: FID-50k feature extraction (reference src/metrics/features.py:17-65) ------------------------
    fid = None
    if args.fid_samples > 0 and args.workload == "biggan128":
        from studiogan_amd import metrics as M
        from studiogan_amd.worker import make_GAN_untrainable
        make_GAN_untrainable(G, w.Gen_ema, D)
        per_rank = (args.fid_samples + world - 1) // world
        fid = {}
        for name, idt, fmode in (("f32", torch.float32, "exact"), ("f32_bf16x3", torch.float32, "bf16x3"), ("bf16", torch.bfloat16, "exact")):
            model = M.LoadEvalModel(device=device, state_dict=M.synthetic_state_dict(0), dtype=idt, f32_mode=fmode)   # seeded random Inception weights (the real ones need network access)
            M.generate_images_and_stack_features(w.Gen_ema, model, 2 * args.batch, args.batch, wl["z_dim"], wl["classes"], device=device)  # warm-up
            barrier()
            mom = M.FeatureMoments(2048, device)
            t1 = time.perf_counter()
            feats, probs, _ = M.generate_images_and_stack_features(w.Gen_ema, model, per_rank, args.batch, wl["z_dim"], wl["classes"], quantize=True,
                                                                   device=device, moments=mom)
            mu, sigma = mom.finalize(group)
            barrier()
            dt_f = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dt_f], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_f = float(t.item())
            fid[name] = {"samples_per_sec": round(per_rank * world / dt_f, 1), "seconds": round(dt_f, 2)}
            del model, feats, probs
        fid = {"metric": "FID-50k feature-extract samples/sec", "samples": per_rank * world, "batch": args.batch,
               "value": fid["f32"]["samples_per_sec"], "unit": "samples/sec", "inception_f32": fid["f32"], "inception_bf16": fid["bf16"],
               # fp32 tensors, every Inception convolution's operands split into two bf16 terms in registers and contracted with three bf16 MFMAs per k-tile (fp32
               # accumulation; functional.f32_mode, csrc/gemm_core.h SPLIT): pool3 features within 6e-6 of the fp32 oracle (tests/test_eval_gpu.py, bound 2e-4 as for
               # the exact path). NOT the headline value: `value` stays the exact-fp32-MFMA run.
               "inception_f32_bf16x3": dict(fid["f32_bf16x3"], note="fp32 storage, bf16x3 split-precision MFMA arithmetic (~2^-16 per product): not the headline value"),
               "includes": "G_ema forward (bf16) + on-device quantize/resize + InceptionV3 + softmax + fp64 moment accumulation",
               # algorithmic work per sample: G forward 42.24 GFLOP + InceptionV3 at 299^2 11.4 GFLOP
               "roofline_bf16": {"bound": "mfma", "gflop_per_sample": 53.64, "achieved": round(fid["bf16"]["samples_per_sec"] * 53.64 / 1e3, 1), "unit": "TFLOP/s",
                                 "peak": 2500.0, "frac": round(fid["bf16"]["samples_per_sec"] * 53.64 / 1e3 / 2500.0, 4),
                                 "kernel_trace": "profiles/r06_fid_leg_f32_exact_kerneltrace.txt / r06_fid_leg_f32_bf16x3_kerneltrace.txt (tools/fid_leg.py under rocprofv3 --kernel-trace --stats)"},
               # the headline value's own roofline: G_ema runs in bf16 (42.24 GFLOP per sample on the bf16 MFMA), InceptionV3 in fp32 (11.4 GFLOP per sample on the
               # fp32 MFMA, 157.3 TFLOP/s): the time each half would need at its peak, summed, over the measured time per sample
               "roofline_f32": {"bound": "mfma", "gflop_per_sample_bf16_generator": 42.24, "gflop_per_sample_f32_inception": 11.4,
                                "inception_f32_tflops_if_it_took_all_the_time": round(fid["f32"]["samples_per_sec"] * 11.4 / 1e3, 1), "peak_f32": 157.3,
                                "frac": round(fid["f32"]["samples_per_sec"] * (42.24 / 2500.0e3 + 11.4 / 157.3e3), 4),
                                "frac_note": "sum over the two halves of (FLOPs per sample / that dtype's MFMA peak) / measured seconds per sample"},
               "weights": "seeded random (pretrained FID Inception weights are not available offline)"}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    failed = []
    ms_per_step = 1e3 * elapsed / args.steps
    global_batch = args.batch * world
    value = global_batch * args.steps / elapsed
    prof = [prof4[(i // 3) * 4 + (i % 3)] for i in range(9)]          # {launches, ms, algorithmic flops} per kind of the engine, over psteps steps
    conv_exec_flop = prof4[3] + prof4[7]
    n_launch = prof[0] + prof[3]
    conv_ms = prof[1] + prof[4]
    conv_flop = prof[2] + prof[5]
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = PEAK_BF16_TFLOPS if mixed else PEAK_F32_TFLOPS
    traffic, traffic_src = None, None
    try:
        if pmc_src and args.workload == "biggan128" and mixed and args.batch == 256:
            traffic = pj["hbm_bytes_per_launch"]
            traffic_src = f"{pmc_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, read side x2)"
    except Exception:
        pass
    # the dominant kernel: sg_conv_q_kernel with and without the fused skip are ONE kernel template (two profiler tags)
    dominant = None
    if per_kernel:
        # which family is the largest: the full table (profiled steps); its figures: the timed region's event pairs when that family is the one bracketed there
        fam_full = {}
        for k, v in per_kernel.items():
            fam_full.setdefault(k.split("<")[0].strip(), []).append(v)
        dname_full = max(fam_full.items(), key=lambda kv: sum(r["ms_per_step"] for r in kv[1]))[0]
        dom_src = per_kernel_timed if any(k.split("<")[0].strip() == dname_full for k in per_kernel_timed) else per_kernel
        fam = {}
        for k, v in dom_src.items():
            fam.setdefault(k.split("<")[0].strip(), []).append(v)
        dname, rows = max(fam.items(), key=lambda kv: sum(r["ms_per_step"] for r in kv[1]))
        dms = sum(r["ms_per_step"] for r in rows)
        dn = sum(r["launches_per_step"] for r in rows)
        dalg = sum((r["tflops"] or 0.0) * r["ms_per_step"] for r in rows) / dms if dms > 0 else 0.0
        dex = sum((r["executed_tflops"] or 0.0) * r["ms_per_step"] for r in rows) / dms if dms > 0 else 0.0
        dab = sum(r["algorithmic_MB_per_launch"] * r["launches_per_step"] for r in rows)
        dpb = sum(r.get("pmc_MB_per_launch", 0.0) * r["launches_per_step"] for r in rows) if all("pmc_MB_per_launch" in r for r in rows) else None
        dominant = {"kernel": dname + (" (plain + fused-skip instantiations)" if len(rows) > 1 else ""),
                    "measured_in_timed_region": dom_src is per_kernel_timed, "ms_per_step": round(dms, 3), "launches_per_step": round(dn, 1),
                    "frac": round(dalg / peak_for(mixed), 4), "executed_frac": round(dex / peak_for(mixed), 4), "tflops": round(dalg, 1), "executed_tflops": round(dex, 1),
                    "algorithmic_MB_per_launch": round(dab / dn, 1) if dn else None,
                    "pmc_bytes_over_algorithmic": round(dpb / dab, 3) if (dpb and dab) else None}
    out = {
        "metric": "images/sec (G+D step) BigGAN ImageNet-128 bs256",
        "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "bf16" if mixed else "f32", "data": "synthetic",
        "data_note": f"{len(pool)} distinct real baskets per rank, cycled: uint8-grid samples of a frozen copy of the initial generator "
                     "(keeps the hinge active: d_loss > 1e-2 is asserted on the last timed step); z / labels drawn on the device every update",
        "last_step_losses": {"d_loss": round(d_last, 5), "g_loss": round(g_last, 5)},
        "host_lead": host_lead,
        "config": {"workload": wl["desc"], "per_gpu_batch": args.batch, "global_batch": global_batch, "d_updates_per_step": wl["n_d"],
                   "parallelism": f"dp{world}" + (" (RCCL grad all-reduce + sync-BN)" if world > 1 else ""),
                   "exchange": None if world == 1 else ("libsgamd sg_allreduce_flat (native RCCL communicator)" if rccl_ranks else
                                                        ("gloo (one-device plumbing run)" if one_dev else "torch.distributed nccl backend (= RCCL)")),
                   "rccl_ranks": rccl_ranks, "sync_bn_exchange": bn_exchange, "comm_notes": comm_notes or None,
                   # time the compute stream spent waiting for collectives (sync-BN all-reduces run on it; waits on the gradient reductions in
                   # FusedAdam.step), max over ranks: hipEvent pairs around every such point (studiogan_amd.comm.exposed)
                   "exposed_comm_ms_per_step": None if exposed_comm_ms is None else round(exposed_comm_ms / args.steps, 3)},
        # frac = SURVEY 8(d)'s accounting (ALGORITHMIC FLOPs of the reference's op graph / time); executed_frac = the MFMAs actually issued / time (the
        # hardware-utilisation figure: the convolutions next to a 2x resampling run through the exact pooled / phase-filter identity of csrc/conv_q.h and
        # execute 16/36 of the algorithmic MACs). Both lead the object; every per-kernel row carries both.
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "executed_tflops": round(conv_exec_flop / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else None,
                     "executed_frac": round(conv_exec_flop / (conv_ms * 1e-3) / 1e12 / peak, 4) if conv_ms > 0 else None,
                     "frac_basis": "frac: algorithmic FLOPs of the reference's op graph (3x3 convolutions over the fine grid) / hipEvent time = SURVEY 8(d); "
                                   "executed_frac: MFMAs issued / the same time = matrix-pipe utilisation (quad launches execute 16/36 of the algorithmic MACs)",
                     # what this chip sustains on the SAME loop with every memory operation removed (profiles/r06_conv_v4_ablation_u.txt, SG_V4_ABLATE=31: twelve waves per
                     # CU issuing v_mfma_f32_32x32x16_bf16 from registers on real activations, no LDS reads, no DMA, no barrier, no epilogue): 1430-1620 TFLOP/s -- the
                     # clocks the power limit leaves under sustained matrix load. A measured context figure, never the denominator of frac.
                     "mfma_only_loop_tflops_measured": [1430, 1620],
                     "executed_frac_of_mfma_only_loop": round(conv_exec_flop / (conv_ms * 1e-3) / 1e12 / 1525.0, 4) if conv_ms > 0 else None,
                     "dominant_kernel": dominant,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                     "pmc_csrc_sha16": pmc_sha, "csrc_sha16": cur_sha,
                     "pmc_stale": None if pmc_src is None else (pmc_sha != cur_sha),
                     "pmc_note": "traffic and every pmc_* figure come from the committed PMC summary named in traffic_source (counters need their own rocprofv3 passes); "
                                 "pmc_stale = that summary was taken on other kernel sources than this run's (csrc hashes differ or the summary predates the hash)",
                     "kernel": "convolution engine (family; per_kernel has the members, dominant_kernel the largest): sg_conv_q_kernel / sg_wgrad_ql_kernel (3x3 next to a 2x resampling as 4x4-stride-2 / four 2x2 phase convolutions) / sg_conv_v4_kernel (3x3 halo, <= 384 channels, G tails with the 1x1 skip fused) / sg_conv_v3_kernel (3x3 halo, deep layers) / sg_conv_sk_kernel (1x1, stem) / sg_conv_rs_kernel (RGB layers) / sg_conv_v2_kernel / sg_wgrad_v3l_kernel / sg_wgrad_sk_kernel / sg_wgrad_v2_kernel / sg_gemm_kernel",
                     "per_kernel": per_kernel, "per_kernel_pmc_source": pmc_src,
                     "launches_per_step": round(n_launch / psteps, 1), "avg_launch_ms": round(conv_ms / max(n_launch, 1), 4),
                     "algorithmic_gflop_per_launch": round(conv_flop / max(n_launch, 1) / 1e9, 3),
                     "algorithmic_bytes_per_launch": round(sum(v["algorithmic_MB_per_launch"] * 1e6 * v["launches_per_step"] for v in per_kernel.values()) /
                                                           max(sum(v["launches_per_step"] for v in per_kernel.values()), 1e-9)) if per_kernel else None,
                     "flop_count_note": "2*I*J*K on the launched (padded) dims: RGB layers run with 8 padded channels, < 1 % above the unpadded count over the step",
                     "conv_ms_per_step": round(conv_ms / psteps, 2),
                     "gemm_ms_per_step": round(prof[7] / psteps, 2),
                     "measured_on": ("event pairs around every engine launch of the timed region (SG_BENCH_PROF_ALL=1)" if prof_all else
                                     f"dominant_kernel: hipEvent pairs around every launch of that kernel IN the timed region ({args.steps} steps). Family total, per_kernel and "
                                     f"roofline_hbm: event pairs around every launch of {psteps} further steps run right after the timed region -- bracketing every engine "
                                     "launch inside the timed region costs ~3 ms of dispatch gaps per step (two barrier packets per launch; tools/kt_gaps.py, "
                                     "profiles/r06_step_gaps_*.txt), i.e. the instrument slowed `value`")},
    }
    if hbm is not None:
        out["roofline_hbm"] = hbm
    if dfwd is not None:
        out["d_forward_stack"] = dfwd
    if fid is not None:
        out["fid_extract"] = fid
    if wl["gflop"]:
        out["step_tflops"] = round(wl["gflop"] * global_batch / 1e3 / (ms_per_step * 1e-3), 2)  # whole-step algorithmic TFLOP/s
    if world == 1 and not args.no_extras and args.workload == "biggan128":
        del w, G, D, pool
        torch.cuda.empty_cache()
        out["extra_workloads"] = {}
        for name in EXTRAS:
            try:
                out["extra_workloads"][name] = run_extra(name, device)
            except Exception as ex:      # an extra never takes the headline line down, but it is counted and (--strict) fails the run
                out["extra_workloads"][name] = {"error": repr(ex)[:300]}
                failed.append(name)
                sys.stderr.write(f"[bench] EXTRA WORKLOAD FAILED: {name}: {ex!r}\n")
        w = G = D = pool = None
    if not args.no_cpu_baseline and world == 1:
        w = G = D = pool = None
        torch.cuda.empty_cache()
        sys.stderr.write("[bench] gpu result: " + json.dumps(out) + "\n")
        out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_batch)
        # C1 / C2 at their FULL batch, and the Inception forward at B = 32, next to their GPU legs
        for name, bound in (("dcgan32_bs64_fp32", 90), ("sngan32_bs256_fp32", 170)):
            if isinstance(out.get("extra_workloads", {}).get(name), dict) and "error" not in out["extra_workloads"][name]:
                out["extra_workloads"][name]["cpu_baseline"] = cpu_baseline(name, 0, bound)
        if fid is not None:
            out["fid_extract"]["cpu_baseline_inception"] = cpu_baseline("inception", 32, 120, "samples/sec")
    out["failed_legs"] = failed
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    if failed and args.strict:
        sys.exit(3)


if __name__ == "__main__":
    main()
