"""GPU: parity of the BENCHMARKED configurations at their real channel widths and resolutions (tests/golden/*w: BigGAN-128 ch 96 =
configs/ImageNet/BigGAN-256.yaml, SNGAN-32 ch 64, WGAN-GP ResNet-128, BigGAN-deep-128 ch 128), with the tile-count / problem-size
heuristics of the dispatchers switched off (SG_CONV_V3 / SG_CONV_V2 / SG_CONV_SK / SG_CONV_RS = force, SG_WGRAD_BJ256 = force) so that the
kernels bench.py runs at batch 256 -- halo conv_v3 (incl. the 96x512 tile), streaming conv_sk, row-streaming conv_rs (RGB layers), conv_v2, wgrad_v2 (both cout tiles),
fused attention at HW = 4096 -- are the kernels under test at the fixtures' small batch.

Three comparisons per fixture (the width-8 fixtures get the same three in test_model_gpu.py / test_blocks_gpu.py):
  * one full training step against the golden vectors written by the REAL reference (fp32 and bf16),
  * the same step in fp32 against the CPU oracle, re-synchronised after every update (tight per-update bound),
  * bf16 forward / backward of D and G against the bf16-emulating oracle (forward <= 2e-2 relative-L2; gradient relative-L2 reported).
fp32 tolerances = base (1e-3 forward / state, 1e-2 gradient relative-L2) + 4 x the ORACLE's OWN movement of that tensor under a 2e-6 relative
weight perturbation (tests/golden/<name>.cond.npz, oracle/make_golden.py conditioning()): ill-conditioned quantities (WGAN-GP critic at
batch 2, generator gradients through every ReLU of D) get a measured bound instead of a hand-picked one.
The measured errors are written to gpurun_out/fullwidth_parity.txt when that directory is writable."""
import os

import pytest
import torch

from test_model_gpu import step_vs_golden, stagewise_vs_oracle
from test_blocks_gpu import bf16_vs_emulating_oracle

pytestmark = pytest.mark.gpu

# The driver gives `pytest -m gpu` 1200 s; round 3's suite took 936 s, 7 min of it CPU-oracle time at full width. The comparisons that add no
# kernel coverage of their own -- the batch curve of the bf16 noise-floor study, the 256 x 256 BigGAN-deep fixture (same kernels as the
# 128 x 128 one except the unfused attention fallback), the stage-wise fp32 runs of the two configurations that are not the benchmarked one --
# run only with SG_SLOW=1 (tools/sessions/r4_slow.sh; their summary is committed under profiles/). Round 5: the step of the 256 x 256 fixture against the
# reference's golden vectors (both dtypes) is back in the default suite -- its D attention (16384 queries x 4096 keys) runs on the streaming-keys kernels now --
# paid for by caching the oracle's own noise floors (tests/make_floors.py -> tests/golden/*.floors.json: one oracle pass instead of two or three per comparison).
slow = pytest.mark.skipif(os.environ.get("SG_SLOW") != "1", reason="slow full-width comparison: run with SG_SLOW=1 (tools/sessions/r4_slow.sh)")

WIDE = ["biggan128w", "sngan32w", "wgangp128w", "bigdeep128w"]
# C4 at the resolution BASELINE.json names (BigGAN-Deep ImageNet-256: reference src/models/big_resnet_deep_legacy.py:80-95 "256" tables,
# attention in D at 128^2 = 16384 positions), batch 2: the step against the reference's golden vectors and stage-wise against the oracle
WIDE256 = ["bigdeep256w"]


@pytest.fixture
def forced(monkeypatch):
    for k in ("SG_CONV_V4", "SG_CONV_V3", "SG_CONV_V2", "SG_CONV_SK", "SG_CONV_RS", "SG_CONV_RS96", "SG_WGRAD_BJ256", "SG_WGRAD_V3"):
        monkeypatch.setenv(k, "force")


def _dump(tag, rows):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fullwidth_parity.txt"), "a") as f:
            worst = {}
            for n, e, t in rows:
                fam = n.split(" ")[0] + " " + (n.split(" ")[1] if " " in n else "")
                fam = fam.split(".")[0]
                if e > worst.get(fam, (0, 0))[0]:
                    worst[fam] = (e, t)
            for fam, (e, t) in sorted(worst.items()):
                f.write(f"{tag:40s} {fam:40s} worst err {e:.3e} (tol {t:.1e})\n")
            for n, e, t in rows:
                if "WHOLE-NETWORK" in n:
                    f.write(f"{tag:40s} {n:40s}       err {e:.3e} (tol {t:.1e})\n")
            if "batch" in tag:          # the batch curve: every tensor, so that what does not average out can be localised per layer
                for n, e, t in rows:
                    if " grad " in n and "WHOLE" not in n:
                        f.write(f"    {tag:36s} {n:60s} err {e:.3e}\n")
    except OSError:
        pass


# (the fp32 step of C4 at 128^2 runs with SG_SLOW=1 only: its kernels and its block code are those of the 256^2 fixture, whose both dtypes are in the default suite)
@pytest.mark.parametrize("name,mixed", [(n, m) for n in WIDE + WIDE256 for m in (False, True) if (n, m) != ("bigdeep128w", False)] +
                         [pytest.param("bigdeep128w", False, marks=slow)])     # (C4 at 256^2 -- its discriminator attends over 16384 positions: the streaming-keys attention kernels -- is in the default suite since round 5)
def test_fullwidth_step_vs_golden(sg, forced, name, mixed):
    step_vs_golden(name, mixed)


# (biggan128w: 58 s of fp32 CPU oracle at full width -- SG_SLOW=1 since round 5; the default suite keeps its fp32 step against the REAL reference's golden vectors
# and the teacher-forced bf16 comparisons, and the driver's 1200 s limit keeps its margin on a slow box)
@pytest.mark.parametrize("name", ["sngan32w"] + [pytest.param(n, marks=slow) for n in ["biggan128w", "wgangp128w", "bigdeep128w"] + WIDE256])
def test_fullwidth_step_stagewise_vs_oracle(sg, forced, name):
    rows = []
    try:
        stagewise_vs_oracle(name, t=1e-3, report=rows)
    finally:
        _dump("stagewise fp32 " + name, rows)


@pytest.mark.parametrize("which", ["D", "G"])
@pytest.mark.parametrize("name", WIDE)     # every network bench.py times in bf16 (C3, C4 at 128^2, C5) and C2
def test_fullwidth_bf16_vs_emulating_oracle(sg, forced, name, which):
    rows = []
    try:
        bf16_vs_emulating_oracle(name, which, report=rows)
    finally:
        _dump(f"bf16-emu {which} " + name, rows)


@pytest.mark.parametrize("name,which,base", [("sngan32w", "G", 6e-2), ("biggan128w", "D", 1e-2), ("biggan128w", "G", 6e-2)])
def test_fullwidth_bf16_teacher_forced_vs_reference_graph_rounding(sg, forced, name, which, base):
    """The quad kernels against the REFERENCE graph's storage points: the emulating oracle with quad_emu=False rounds each 3x3 filter
    entry once, as autocast does for the reference, while csrc/conv_q.h rounds the summed phase filter once more -- so this comparison is NOT against a model of the
    kernel's own rounding. Teacher-forced, so nothing compounds: every block output, block-input gradient and weight gradient of C2's generator (three quad layers) and
    of C3's discriminator and generator (biggan128w = configs/ImageNet/BigGAN-256.yaml at full width: five pooling blocks resp. five upsampling blocks, attention)
    within `base` relative-L2. Measured on the MI355X (profiles/r06_reference_rounding_table.txt, per tensor): sngan32w G 3.7e-2 worst (the quad layer's own weight
    gradient; 2.1e-2 upstream of it), against 9e-4 when the oracle restates the second rounding; biggan128w D 2.8e-3 worst teacher-forced (bound 1e-2), its logits
    9.2e-4 and pooled features 7.9e-4 of the reference graph's; biggan128w G 3.8e-2 worst teacher-forced, image 1.8e-2 (SURVEY 8c's 2e-2)."""
    rows = []
    try:
        bf16_vs_emulating_oracle(name, which, report=rows, quad_emu=False, teacher_base=base)
    finally:
        _dump(f"bf16-emu {which} {name} reference-graph rounding", rows)
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        try:
            with open(os.path.join(d, "reference_rounding_table.txt"), "a") as f:
                for n, e, t in rows:
                    f.write(f"{name:12s} {which} {n:70s} err {e:.3e} (tol {t:.1e})\n")
        except OSError:
            pass


@pytest.mark.parametrize("name,gscale", [("sngan32w", 1.0), ("sngan32", 8.0), ("dcgan32", 4.0)])
def test_fp32_training_step_in_bf16x3_mode_vs_golden(sg, forced, name, gscale):
    """The fp32 training step with every convolution of the generic engine -- forward, data gradient, weight gradient -- in the "bf16x3" arithmetic (fp32 tensors,
    operands split into two bf16 terms in registers, three bf16 MFMAs per k-tile; functional.f32_mode, csrc/gemm_core.h SPLIT) against the REAL reference's fp32
    golden vectors. C2's network at full width (sngan32w): at the EXACT path's tolerances (2e-4 forward, 1e-3 gradients + the oracle's measured conditioning).
    The forward bounds stay the exact path's everywhere; the gradient bounds of the width-8 fixture are x8 and of the full-width DCGAN x4 (measured on the MI355X,
    profiles/r06_pytest_split_train_o.txt: 6.2e-3 against 1e-3 resp. 8.8e-3 against 3e-3 worst): the mode carries ~5e-6 of forward rounding noise instead of ~1e-6, which puts that many more
    ReLU inputs on the other side of zero -- the tie effect of DESIGN.md 3, not an arithmetic error of the gradient kernels (kernel level: 4e-6, test_kernels_gpu.py)."""
    from studiogan_amd import functional as F, _lib as L
    with F.f32_mode("bf16x3"):
        step_vs_golden(name, False, gscale=gscale)
    assert L.lib().sg_get_f32_mode() == 0


# bf16 weight-gradient agreement with the emulating oracle as a function of the batch. Result (profiles/r03_bf16_batch_curve.txt):
# D: whole-network gradient 1.1 % at batch 4 -> 1.0 % at batch 32, oracle's own floor 0.6 %: held to 5 %. G: 16.9 % -> 16.5 %, flat -- and so is
# the ORACLE'S OWN movement under a 1e-5 weight perturbation (18.8 % -> 18.2 %): with a random linear functional as the objective the weight gradient
# is a random-walk sum over pixels (signal ~ sqrt(N)), the units whose ReLU mask flips under rounding noise contribute ~ sqrt(f N), and the ratio
# sqrt(f) per layer is independent of the batch; the error grows layer by layer from the output (conv2d5 1.6 %, block 5 9-13 %, block 0 17 %).
# With ONE upstream-gradient image shared by all samples (a partly coherent signal ~ N) the oracle's floor falls 16.7 % -> 9.9 % from batch 4 to 32,
# as does the HIP path's distance. Every comparison is bounded by max(base, 1.5 x the oracle's own movement of that tensor), measured in the test.
CURVE = [("biggan128w", "D", False, 0.05), ("biggan128w", "G", False, 0.10), ("biggan128w", "G", True, 0.10)]


@slow
@pytest.mark.parametrize("batch", [8, 32])
@pytest.mark.parametrize("name,which,shared,base", CURVE)
def test_fullwidth_bf16_gradient_batch_curve(sg, forced, name, which, shared, base, batch):
    rows = []
    try:
        whole, floor = bf16_vs_emulating_oracle(name, which, report=rows, batch=batch, tg=base, floor=True, shared_objective=shared)
        if which == "D":
            assert whole <= 0.05, f"D whole-network bf16 weight gradient at batch {batch}: {whole:.3e}"
    finally:
        _dump(f"bf16-emu batch {batch:3d} {which}{' shared-objective' if shared else ''} " + name, rows)
