"""GPU: the differentiable augmentations and the l2 loss (csrc/ext/augment.hip behind studiogan_amd.diffaug / .cr / losses.l2_loss)
  * against the vectors the reference's own apply_diffaug / apply_cr_aug / MSELoss wrote under a seeded generator (tests/golden/aug.npz): outputs
    (bit for bit where no contrast mean enters), gradients and the second-order term an R1 penalty through an augmented batch needs;
  * every operator subset against oracle/aug_ref.py (pinned bit-identically to the reference) in the kernel's fixed order, both translation kinds;
  * at the benchmark's batch (256 x 3 x 128 x 128) through size-independent properties: adjointness of the backward kernel to the forward one, linearity,
    the brightness offset, run-to-run bit-identity, and the seeded device draws of the host mirror against the oracle fed with the same draws.
checks shared with the CPU-interpreter run of the same kernel sources: tests/aug_checks.py."""
import pytest
import torch

import aug_checks as AC
from oracle import aug_ref as AR

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("case", AC.DIFFAUG_CASES, ids=[c[0] for c in AC.DIFFAUG_CASES])
def test_diffaug_matches_reference_vectors(sg, case):
    AC.diffaug_case(case, DEV)


@pytest.mark.parametrize("case", AC.CR_CASES, ids=[c[0] for c in AC.CR_CASES])
def test_cr_aug_matches_reference_vectors(sg, case):
    AC.cr_case(case, DEV)


@pytest.mark.parametrize("case", AC.MSE_CASES, ids=[c[0] for c in AC.MSE_CASES])
def test_l2_loss_matches_reference_vectors(sg, case):
    AC.mse_case(case, DEV)


@pytest.mark.parametrize("kind", AC.LOSS_KINDS)
def test_least_squares_and_logistic_losses_match_reference_vectors(sg, kind):
    AC.loss_case(kind, DEV)


@pytest.mark.parametrize("case", AC.FM_CASES, ids=[c[0] for c in AC.FM_CASES])
def test_feature_matching_loss_matches_reference_vectors(sg, case):
    AC.fm_case(case, DEV)


def test_apa_select_and_weight_clipping(sg):
    for i in range(3):
        AC.apa_case(i, DEV)
    AC.clamp_case(DEV)


def test_augment_operator_subsets_vs_oracle(sg):
    from studiogan_amd import _lib as L
    every = [L.AUG_BRIGHTNESS, L.AUG_SATURATION, L.AUG_CONTRAST, L.AUG_FLIP, L.AUG_CUTOUT]
    k = 0
    for tr in (0, L.AUG_TRANSLATE, L.AUG_TRANSLATE_REFLECT):
        for sub in range(1 << len(every)):
            ops = tr | sum(b for i, b in enumerate(every) if sub >> i & 1)
            if ops:
                AC.oracle_spec_case(((2, 3, 8, 12), (3, 3, 9, 7), (2, 1, 16, 16), (2, 4, 6, 10))[k % 4], ops, DEV, seed=k)
            k += 1


def test_augment_properties_at_benchmark_size(sg):
    from studiogan_amd import _lib as L
    allz = L.AUG_BRIGHTNESS | L.AUG_SATURATION | L.AUG_CONTRAST | L.AUG_FLIP | L.AUG_CUTOUT
    a = AC.adjoint_and_linearity((256, 3, 128, 128), allz | L.AUG_TRANSLATE, DEV, 1)
    b = AC.adjoint_and_linearity((256, 3, 128, 128), allz | L.AUG_TRANSLATE, DEV, 1)
    assert torch.equal(a, b)                                   # fixed-order sums: bit-identical between runs
    AC.adjoint_and_linearity((64, 3, 256, 256), allz | L.AUG_TRANSLATE_REFLECT, DEV, 2)
    AC.adjoint_and_linearity((256, 3, 32, 32), L.AUG_SATURATION | L.AUG_CONTRAST | L.AUG_TRANSLATE_REFLECT, DEV, 3)


def test_diffaug_seeded_device_draws_vs_oracle(sg):
    """no replay: the host mirror draws on the device; the same seed replayed through the reference's own draw calls (oracle/aug_ref.draw_diffaug on the
    device) gives the oracle the same numbers -- the mirror consumes the generator exactly like the reference (count, order, shapes, dtypes)"""
    from studiogan_amd import diffaug as DA, cr as CR
    shape = (16, 3, 64, 64)
    x = (torch.rand(shape, generator=torch.Generator().manual_seed(3)) * 2 - 1)
    for policy in ("color,translation,cutout", "translation,cutout", "cutout,color,translation"):
        torch.manual_seed(11)
        y = DA.apply_diffaug(x.to(DEV), policy)
        after = torch.rand(3, device=DEV)
        torch.manual_seed(11)
        draws = [d.cpu() for d in AR.draw_diffaug(shape, policy, device=DEV)]
        assert torch.equal(after, torch.rand(3, device=DEV)), policy
        AC.check(f"diffaug {policy} device draws", y, AR.diffaug(x, policy, draws), AC.TOL)
    torch.manual_seed(12)
    y = CR.apply_cr_aug(x.to(DEV))
    torch.manual_seed(12)
    coin, tx, ty = AR.draw_cr(shape, device=DEV)
    assert torch.equal(y.cpu(), AR.cr_aug(x, coin, tx.cpu(), ty.cpu()))
    with pytest.raises(KeyError):
        DA.apply_diffaug(x.to(DEV), policy="colour")           # diffaug.py:40: an unknown policy entry is a KeyError


def test_augment_argument_errors(sg):
    from studiogan_amd import functional as F, _lib as L, losses
    x = torch.randn(2, 3, 8, 8, device=DEV)
    geom = torch.zeros(2, 5, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="exclusive"):
        F.AugmentFn.apply(x, F.AugSpec(L.AUG_TRANSLATE | L.AUG_TRANSLATE_REFLECT, None, geom))
    with pytest.raises(RuntimeError, match="colour table"):
        F.AugmentFn.apply(x, F.AugSpec(L.AUG_CONTRAST, None, None))
    with pytest.raises(RuntimeError, match="geometry table"):
        F.AugmentFn.apply(x, F.AugSpec(L.AUG_CUTOUT, None, None, 4, 4))
    with pytest.raises(RuntimeError, match="max_t"):
        F.AugmentFn.apply(x, F.AugSpec(L.AUG_TRANSLATE_REFLECT, None, geom, 0, 0, 8))
    with pytest.raises(RuntimeError, match="fp32"):
        F.AugmentFn.apply(x.bfloat16(), F.AugSpec(L.AUG_FLIP, None, geom))
    with pytest.raises(RuntimeError, match=r"\[2, 5\]"):
        F.AugmentFn.apply(x, F.AugSpec(L.AUG_FLIP, None, geom[:1]))
    with pytest.raises(RuntimeError, match="shapes differ"):
        losses.l2_loss(x, x[:1])
    big = torch.randn(5_000_000, device=DEV)                   # more elements than one pass of the partial-sum grid
    ref = float(((big.double() - 0.25) ** 2).mean())
    assert abs(float(losses.l2_loss(big, torch.full_like(big, 0.25))) - ref) <= 2e-6 * ref


@pytest.mark.parametrize("tag", AC.CONSISTENCY_CASES)
def test_worker_update_with_diffaug_and_consistency_regularisers(sg, tag):
    """studiogan_amd.worker.Worker with apply_diffaug / apply_cr / apply_bcr / apply_zcr: the loss and every parameter gradient of one discriminator and
    one generator update against the REAL reference's models + utils/diffaug.py + utils/cr.py + MSELoss combined as src/worker.py:236-365,520-603
    (tests/golden/consistency.npz)"""
    AC.consistency_case(tag, DEV)


@pytest.mark.parametrize("name", ["biggan32", "sngan32"])
def test_r1_through_diffaug(sg, name):
    AC.r1_through_diffaug_case(name, DEV)


def test_ada_pipeline_matches_reference_vectors(sg):
    """studiogan_amd.ada_aug.AdaAugment ('blit', 'geom', 'color', 'bgc'; RGB and one-channel images) fed the draws the REAL reference's AdaAugment made:
    output and image gradient (tests/golden/ada.npz)"""
    from oracle import make_golden_ada as MGD
    for case in MGD.CASES[:7]:      # (the image-space filtering / noise / cutout cases: tests/test_wide_ada_gpu.py)
        AC.ada_case(case, DEV)


def test_ada_operators_adjoint_at_benchmark_size(sg):
    AC.ada_adjoint_case((64, 3, 128, 128), DEV, 1)
    AC.ada_adjoint_case((256, 3, 32, 32), DEV, 2)
    AC.ada_adjoint_case((8, 1, 33, 47), DEV, 3)
    from studiogan_amd import ada_aug
    aug = ada_aug.AdaAugment(**ada_aug.AUGPIPE["bgc"]).to(DEV)
    aug.p.copy_(torch.as_tensor(0.6))
    x = torch.rand(256, 3, 128, 128, device=DEV) * 2 - 1
    torch.manual_seed(5)
    a = aug(x)
    torch.manual_seed(5)
    b = aug(x)
    assert a.shape == x.shape and torch.equal(a, b) and bool(torch.isfinite(a).all())
